"""A/B of library variants on one box, interleaved so that drift hits all of them alike:
    python3 profiles/ab_libs.py [rounds] lib_a.so lib_b.so ...   (paths relative to the repo root)
Each sample is its own process (the library is chosen at import)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(sys.argv[1])
libs = sys.argv[2:]
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ)
        spec = lib.split(":")
        env["ICP_MI355X_LIB"] = os.path.join(ROOT, spec[0])
        for kv in spec[1:]:
            k, v = kv.split("=")
            env[k] = v
        p = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "ab_one.py")], env=env, capture_output=True,
                           text=True, timeout=300)
        print(f"round {r} {lib:50s} {p.stdout.strip().splitlines()[-1] if p.stdout.strip() else 'FAILED ' + p.stderr[-300:]}", flush=True)
