"""One of the INDEPENDENT processes of tests/test_gpu_tenants.py: its own handle on the one GPU, `calls` registrations
of a frame-sized cloud (a 28k-point window: what examples/scan3d.rs:113-133 runs per frame), each timed.  The processes
share nothing but the device: the one-launch inner loops of one may find the CUs held by the other's kernels, and must
then cost a bounded wait -- never a wrong bit, never a quarter of a second."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import icp_rust_amd as I  # noqa: E402
from icp_rust_amd import synth  # noqa: E402


def main():
    calls, seed_off = int(sys.argv[1]), int(sys.argv[2])
    workload = sys.argv[5] if len(sys.argv) > 5 else "frame"
    if workload == "converging":
        #  a pair whose inner loops apply many updates: EVERY outer iteration is a one-launch inner
        # loop (k_gn_loop: 256 workgroups that wait for each other at grid barriers), in both tenants at once -- the frame
        # below applies one update per iteration and takes the host-stepped bet instead, where nothing contends
        s3, d3 = synth.converging_pair(100_000, 100_000)[:2]
        init = I.Transform()
    else:
        pk = synth.synthetic_scan3d_packets(150)
        s3 = synth.remove_invalid_values(pk[:75])
        d3 = synth.remove_invalid_values(pk[75:150])
        init = I.Transform([0.01 * seed_off, -0.005, 0.0005 * seed_off])  # (each process its own registration)
    icp = I.Icp3d(d3)
    ref, _, ref_inner = icp.estimate(s3, init, 20, return_info=True)  # (first call: allocations)
    # wait for the other tenant to be up (a file each, in the directory the test made)
    sync = sys.argv[3]
    open(os.path.join(sync, f"ready_{os.getpid()}"), "w").close()
    t_end = time.time() + 60
    while len([f for f in os.listdir(sync) if f.startswith("ready_")]) < int(sys.argv[4]) and time.time() < t_end:
        time.sleep(0.005)
    times, same = [], True
    for _ in range(calls):
        t0 = time.perf_counter()
        T, _, inner = icp.estimate(s3, init, 20, return_info=True)
        times.append(time.perf_counter() - t0)
        same = same and np.array_equal(T.as_array(), ref.as_array()) and np.array_equal(inner, ref_inner)
    times.sort()
    total, worst, p98 = sum(times), times[-1], times[min(len(times) - 1, int(0.98 * len(times)))]
    p50 = times[len(times) // 2]
    print(f"tenant {seed_off}: {calls} calls, mean {1e3 * total / calls:.3f} ms, p50 {1e3 * p50:.3f} ms, p98 {1e3 * p98:.3f} ms, worst {1e3 * worst:.3f} ms, same bits every call: {same}, "
          f"loop (launches, evals, handbacks) {I.gn_loop_counters(icp)} timeouts {I.gn_loop_timeouts(icp)} pose {ref.as_array().tolist()}",
          flush=True)
    icp.close()
    sys.exit(0 if same else 3)


if __name__ == "__main__":
    main()
