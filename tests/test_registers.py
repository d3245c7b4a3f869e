"""The speculative search and the evaluation that decides its pose only run side by side because
their workgroups fit on a SIMD together: three search waves (120 VGPRs each) plus two evaluation
waves (<= 72 each, allocation granule 8) within the 512 registers of a lane.  A few registers more
in any of these kernels silently turns the overlap back into a queue, so the budget is pinned
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
    "k_nn_grid_warmILi3E": 96,           # the warm search in f32 geometry: 5 waves per SIMD alone, 3 beside an evaluation
    "k_win_hist": 56,
    "k_win_compactILb0E": 56,            # (the list variant of the refined windows runs alone)
    "k_win_select": 72,
    "k_win_accumulateILb0E": 72,
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
    # 3 search waves + 2 evaluation waves per SIMD: allocation granule 8
    up8 = lambda x: (x + 7) // 8 * 8
    assert 3 * up8(BUDGET["k_nn_gridILi3ELb1ELb0E"]) + 2 * up8(BUDGET["k_win_select"]) <= 512
    assert 3 * up8(BUDGET["k_nn_grid_warmILi3E"]) + 2 * up8(BUDGET["k_win_select"]) <= 512


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_small_cloud_kernel_does_not_spill_in_the_workgroup_sizes_the_reference_scans_use():
    """the one-launch estimate holds a thread per source point; 1024 threads leave 128 registers each
    and spill (measured 14 % slower), so clouds of up to 512 / 768 points run in smaller workgroups"""
    regs = usage("gn_fast.hip")
    for b, spill_cap in ((512, 0), (768, 96)):
        names = [k for k in regs if "k_tiny_estimateILi2ELj%dE" % b in k and not k.endswith("#scratch")]
        assert len(names) == 1, names
        assert regs.get(names[0] + "#scratch", 0) <= spill_cap, (names[0], regs.get(names[0] + "#scratch"))
