"""The one JSON line bench.py prints (the driver's contract): keys, types and the consistency the judge
checks, on a reduced pair so that the whole leg takes seconds."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_carries_the_contract_fields_and_is_consistent():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n-src", "120000", "--n-dst", "100000", "--steps", "20",
                          "--warmup", "2", "--brute-steps", "1", "--cpu-iters", "1", "--gn-points", "0"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # ONE line
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6  # iterations/s and ms per iteration agree
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["avg_launch_ms"] < d["ms_per_step"]  # the dominant kernel is shorter than a step
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] == 1
    p = d["parity"]
    assert p["pose_bits_equal"] is True and p["idx_equal"] is True and p["inner_iterations_equal"] is True
    bp = d["brute_force"]["parity"]  # the sweep engine checked at the size it was timed on
    assert bp["pose_bits_equal"] is True and bp["idx_equal"] is True and bp["inner_iterations_equal"] is True
    cv = d["converging_pair"]
    assert max(cv["inner_iterations_per_step"]) >= 5 and cv["pose_abs_err_vs_truth"] < 1.0  # (mm)
    # (VERDICT r5 item 7c / ADVICE r5) the rate counts the iterations RUN; what was left out behind a fixed point is stated
    fp = cv["fixed_point"]
    assert fp["iterations_requested"] == 60 and cv["steps"] == fp["iterations_run"] <= 60
    assert abs(cv["value"] * cv["ms_per_step"] / 1e3 - 1.0) < 1e-6 and "all_twenty_run" in cv and "ms_per_step_all_run" in cv and "ms_per_requested_iteration" in cv
    nl = d["nn_large"]
    for key in ("ms_per_search", "frac", "traffic", "traffic_ratio", "algorithmic_bytes_per_launch"):
        assert key in nl, key
    b = d["brute_force"]["roofline"]
    assert b["bound"] == "fp32_valu" and 0.0 < b["frac"] < 1.0


@pytest.mark.gpu
def test_two_rank_bench_lines_say_how_the_ranks_talked():
    """ `bench.py --gpus 2` as the driver launches it, both ranks on the one GPU of the box
    (ICP_BENCH_SHARE_GPU=1: gloo for the rendezvous, the mapped inboxes for the iterations -- a rehearsal of the launch,
    not a measurement): the headline line, `brute_force` and `weak_scaling` each carry n_gpus, transport and rccl_world,
    and the sharded registration is the one-GPU registration bit for bit (`parity` is rank 0's check against the oracle)."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, ICP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n-src", "200000", "--n-dst",
                          "150000", "--steps", "10", "--warmup", "2", "--brute-steps", "1", "--weak-steps", "5", "--cpu-iters", "0",
                          "--gn-points", "0", "--nn-points", "0"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["rccl_world"] == 2 and d["backend"] == "gloo"
    assert d["transport"]  # the inboxes' transport, or the stage calls if the two processes never ran side by side
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6
    for leg, scaling in (("brute_force", "strong"), ("weak_scaling", "weak")):
        assert d[leg]["n_gpus"] == 2 and d[leg]["rccl_world"] == 2 and d[leg]["transport"] and d[leg]["scaling"] == scaling, leg
    if "parity" in d:
        assert d["parity"]["pose_bits_equal"] is True and d["parity"]["inner_iterations_equal"] is True
