"""The warm search kernel alone (stage calls on one stream: nothing beside it) on the 1M benchmark pair, for every
environment given on the command line as KEY=VALUE[,KEY=VALUE] groups (one child process each, experiments build):
    python3 profiles/search_probe.py "" ICP_NN_XCD_CHUNK=16 ICP_NN_XCD_CHUNK=64 LIB=i3r4,ICP_NN_WARM_COOP=1
(LIB=NAME: the variant icp_rust_amd/lib/libicp_ab_NAME.so of profiles/build_variant.sh instead of the experiments build)
prints the HIP-event average of the search launches and a short whole-step figure."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child():
    import numpy as np, torch
    import icp_rust_amd as I
    from icp_rust_amd import synth
    from icp_rust_amd.dist import HipStages, ShardedIcp
    src, dst = synth.synthetic_pair(1_000_000, 1_000_000)
    d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
    icp = I.Icp3d(d_dst)
    solo = ShardedIcp(HipStages(icp), len(src), 0, 1)
    T = I.Transform()
    solo.stages.prepare(d_src, T)
    for _ in range(4):
        T, _ = solo.step(d_src, T)
    icp.profile_enable(1); icp.profile_read()
    for _ in range(16):
        T, _ = solo.step(d_src, T)
    ms, nl = icp.profile_read(); icp.profile_enable(0)
    icp2 = I.Icp3d(d_dst)
    for _ in range(2): icp2.estimate(d_src, I.Transform(), 20)
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); icp2.estimate(d_src, I.Transform(), 20); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20)
    ts.sort()
    print(f"search alone {1e3 * ms / max(nl, 1):7.2f} us ({nl} launches); estimate(20): {1e3 * ts[len(ts) // 2]:.4f} ms/step", flush=True)

if __name__ == "__main__":
    if os.environ.get("SEARCH_PROBE_CHILD"):
        child(); sys.exit(0)
    lib = os.path.join(ROOT, "icp_rust_amd", "lib", "libicp_mi355x_exp.so")
    for grp in (sys.argv[1:] or [""]):
        env = dict(os.environ, SEARCH_PROBE_CHILD="1", ICP_MI355X_LIB=lib)
        for kv in filter(None, grp.split(",")):
            k, v = kv.split("=")
            if k == "LIB":  # a variant built by profiles/build_variant.sh
                env["ICP_MI355X_LIB"] = os.path.join(ROOT, "icp_rust_amd", "lib", f"libicp_ab_{v}.so")
            else:
                env[k] = v
        out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("search alone")]
        print(f"{grp or '(default)':40s} {line[-1] if line else 'FAILED ' + out.stderr[-300:]}", flush=True)
