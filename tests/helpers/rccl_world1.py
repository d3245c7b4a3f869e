"""tests/test_gpu_rccl.py: the collectives the sharded driver issues (icp_rust_amd/dist.py: TorchComm, bench.py), on
the RCCL backend with ONE rank -- the only RCCL world a one-GPU box can form (two ranks on one device are refused as
duplicates).  What it shows: RCCL initialises in this image, accepts the dtypes and the foreign-memory tensor views the
driver hands it (the handle's histogram buffer through __cuda_array_interface__), and `all_ok` picks device tensors
for it.  What it cannot show: two devices."""
import datetime
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import icp_rust_amd as I  # noqa: E402
from icp_rust_amd import synth  # noqa: E402
from icp_rust_amd.dist import HipStages, TorchComm  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=120))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    src, dst = synth.synthetic_pair(50_000, 50_000)
    d_src, d_dst = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
    icp = I.Icp3d(d_dst)
    st = HipStages(icp)
    # the handle's histogram buffer as the driver wraps it (a view of library memory, no copy)
    a, b = torch.empty((50_000, 2), dtype=torch.float64, device="cuda"), torch.empty((50_000, 2), dtype=torch.float64, device="cuda")
    T = I.Transform()
    st.prepare(d_src, T)
    st.correspond(d_src, T, a, b)
    rc, hist = st.eval_hist(a, b, 50_000, 0, 1, I.Transform(), 0)
    assert hist is not None and hist.dtype == torch.int32 and hist.is_cuda, (rc, hist)
    before = hist.clone()
    dist.all_reduce(hist, op=dist.ReduceOp.SUM)  # (TorchComm.sum_)
    torch.cuda.synchronize()
    assert torch.equal(hist, before)
    st.eval_abort()
    # TorchComm.gather: bytes and points
    for send in (torch.arange(4096, dtype=torch.uint8, device="cuda"), torch.rand((1000, 4), dtype=torch.float64, device="cuda"),
                 torch.arange(1000, dtype=torch.int32, device="cuda")):
        recv = torch.empty_like(send)
        dist.all_gather_into_tensor(recv.view(-1), send.view(-1))
        torch.cuda.synchronize()
        assert torch.equal(recv, send)
    # the agreement flag (TorchComm.all_ok, forced through its collective branch) and bench.py's max over ranks
    comm = TorchComm(0, 1)
    comm.world = 2  # (only to take the branch that issues the all_reduce; the group has one rank)
    assert comm.all_ok(True) is True and comm.all_ok(False) is False
    t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 1.25
    out = [None]
    dist.all_gather_object(out, ("node", "0000:05:00.0"))
    assert out == [("node", "0000:05:00.0")]
    dist.barrier()
    icp.close()
    dist.destroy_process_group()
    print("rccl world-1: every collective of the sharded driver ran", flush=True)


if __name__ == "__main__":
    main()
