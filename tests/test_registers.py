"""Kernels that share a SIMD only do so while their workgroups fit on it together (512 registers per lane,
allocation granule 8):
  * round 5: the two evaluations of an outer iteration run side by side in ONE launch of 2 x 256 workgroups
    (k_win_hist_sums_bkt2): two workgroups of 512 threads per CU = four waves per SIMD, i.e. at most 128 registers;
  * the second-pass form behind a search (k_win_hist_sums / k_win_finish, wide windows) must leave room for a second
    workgroup per CU -- two evaluations in flight at once otherwise queue behind each other, which cost the 1M
    pair 4 % when k_win_finish needed 152 registers (profiles/r03_eval_fusion_ab.txt);
  * the stage kernels of a sharded evaluation (k_win_compact) run beside three search waves.
A few registers more in any of these kernels silently turns the overlap back into a queue, so the budget is pinned
here (hipcc cross-compiles without a GPU; -Rpass-analysis prints the allocation)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "icp_rust_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

BUDGET = {  # demangled-name fragment -> max VGPRs
    "k_nn_gridILi3ELb1ELb0E": 120,       # 3-D search with the pose applied, f64 geometry (cold calls, extreme cell sizes)
    "k_nn_grid_warm_coopILi3E": 96,      # the warm search (walk shared by the wave): 5 waves per SIMD alone, 3 beside an evaluation
    "k_nn_grid_seededILi3E": 104,        # ... and the first search of a snapshot (seed + the same walk)
    "k_nn_grid_warmILi3ELb1E": 104,      # ... leaving certificates (settled registrations: no speculative overlap then)
    "k_win_compactILb0E": 72,            # sharded stage call (the list variant of the refined windows runs alone)
    "k_win_hist_sumsE": 112,             # the second-pass form (not the deep-batch variant beyond 4M points, which runs alone)
    "k_win_hist_sums_bktE": 128,         # filed candidates: first launch ...
    "k_win_hist_sums_bkt2E": 128,        # ... of two evaluations side by side: two workgroups per CU
    "k_win_pickE": 128,                  # the one-workgroup finish (a workgroup of 512 threads: 128 registers at most)
    "k_win_pick2E": 128,
    "k_win_finish": 120,
}


def usage(src):
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                          "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(CSRC, src), "-o", os.devnull,
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    regs, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"VGPRs: (\d+)", line)
        if m and name:
            regs[name] = int(m.group(1))
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            regs[name + "#scratch"] = int(m.group(1))
    return regs


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_kernels_that_share_a_simd_stay_within_their_register_budget():
    regs = {}
    regs.update(usage("nn_grid.hip"))
    regs.update(usage("gn_win.hip"))
    seen = set()
    for name, v in regs.items():
        for frag, cap in BUDGET.items():
            if frag in name and not name.endswith("#scratch"):
                seen.add(frag)
                assert v <= cap, f"{name}: {v} VGPRs > {cap}"
                assert regs.get(name + "#scratch", 0) == 0, f"{name} spills"
    assert seen == set(BUDGET), sorted(set(BUDGET) - seen)
    up8 = lambda x: (x + 7) // 8 * 8
    # 3 search waves + 2 waves (one workgroup of 512 threads) of a stage kernel / the second-pass form, per SIMD
    for k in ("k_win_compactILb0E", "k_win_hist_sumsE"):
        assert 3 * up8(BUDGET["k_nn_grid_warm_coopILi3E"]) + 2 * up8(BUDGET[k]) <= 512, k
    assert 3 * up8(BUDGET["k_nn_gridILi3ELb1ELb0E"]) + 2 * up8(BUDGET["k_win_compactILb0E"]) <= 512
    # two workgroups per CU of the paired first launch (4 waves per SIMD)
    assert 4 * up8(BUDGET["k_win_hist_sums_bkt2E"]) <= 512
    # two workgroups of the finishing launch per CU (4 waves per SIMD)
    assert 4 * up8(BUDGET["k_win_finish"]) <= 512


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_small_cloud_kernel_does_not_spill_in_the_workgroup_sizes_the_reference_scans_use():
    """the one-launch estimate holds a thread per source point; 1024 threads leave 128 registers each
    and spill (measured 14 % slower), so clouds of up to 512 / 768 points run in smaller workgroups -- which must not
    spill at all (the sums are folded one dimension at a time for that)"""
    regs = usage("gn_fast.hip")
    for b, spill_cap in ((512, 0), (768, 0), (1024, 96)):
        names = [k for k in regs if "k_tiny_estimateILi2ELj%dE" % b in k and not k.endswith("#scratch")]
        assert len(names) == 1, names
        assert regs.get(names[0] + "#scratch", 0) <= spill_cap, (names[0], regs.get(names[0] + "#scratch"))
