"""A/B of builds of the library on the benchmark pair: for every shared object given, one short bench run
in a child process (ICP_MI355X_LIB selects the build); prints iterations/s and the search kernel's duration
in the timed region and alone.  Same box, back to back, two rounds (box-to-box spread is +-1.5 %).

    python3 profiles/nn_variants.py icp_rust_amd/lib/libicp_mi355x.so icp_rust_amd/lib/libicp_var_*.so
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(lib, extra_env=None):
    env = dict(os.environ, ICP_MI355X_LIB=os.path.abspath(lib), **(extra_env or {}))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "5",
                          "--brute-steps", "0", "--cpu-iters", "0", "--gn-points", "0"], env=env, capture_output=True,
                         text=True, timeout=600)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if not line:
        return None
    d = json.loads(line[-1])
    return d["value"], d["roofline"]["avg_launch_ms"] * 1e3, (d["roofline"]["alone"] or {}).get("avg_launch_ms", 0) * 1e3


def main():
    libs = sys.argv[1:]
    envs = [None]
    if os.environ.get("NNV_GRID_SWEEP"):
        envs = [None] + [{"ICP_GRID_OCC": o, "ICP_GRID_FX": f} for o in ("1", "2", "3", "4") for f in ("2", "4", "8")]
    for rnd in range(2):
        for lib in libs:
            for e in envs:
                r = run(lib, e)
                tag = os.path.basename(lib) + (" " + " ".join(f"{k}={v}" for k, v in e.items()) if e else "")
                print(f"round {rnd} {tag:60s} " + (f"{r[0]:8.1f} it/s  search {r[1]:6.1f} us timed, {r[2]:6.1f} us alone" if r else "FAILED"),
                      flush=True)


if __name__ == "__main__":
    main()
