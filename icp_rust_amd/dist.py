"""Multi-GPU driver: one ICP registration across the GPUs of a node, bit-identical to one GPU.

SURVEY.md 8(e); include/icp_mi355x.h section 5; icp_rust_amd/csrc/shard.hip.  The target cloud is
replicated; the SOURCE cloud is sharded by REDUCTION-TREE BLOCK: the N-term sums of the inner loop are
folded in a fixed tree of `blocks x 512` threads (icp_reduce_geometry), and rank r owns the blocks
[blocks r / W, blocks (r + 1) / W) -- i.e. the source points those blocks fold.  It searches THEIR
nearest neighbours and evaluates THEIR residuals, histogram counts and block sums; per evaluation three
small exchanges cross the ranks (integer histograms summed; order-statistic candidates gathered; block
sums gathered in block order), after which every rank holds the totals one GPU would have computed, to
the bit, and takes the same decisions (3x3 solve, break tests, pose update) on its own.  No pair or
index all-gather remains on the steady-state path; the first evaluation of a kind, and an evaluation
whose predicted window missed, fall back to gathering the pairs and evaluating them replicated.

The orchestration is written over a list of LOCAL ranks and a `comm`:
  * one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; gloo in the CPU tests):
    one local rank, `TorchComm`;
  * N "virtual ranks" in one process (N handles, possibly on one GPU -- the 1-GPU test of the N-rank
    path): all ranks local, `LocalComm`.
The compute is delegated to `stages` objects: `HipStages` (the C ABI) in production; the CPU tests
inject an oracle-backed stand-in.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib
from .api import Transform

INNER_MAX_ITER = 200          # src/lib.rs:61
DELTA_NORM_THRESHOLD = 1e-6   # src/lib.rs:60


def _same_bits(T1, T0):
    """two poses equal bit for bit (objects with .as_array(): api.Transform, or the CPU stand-ins of the gloo tests)"""
    a, b = np.ascontiguousarray(T1.as_array(), dtype=np.float64), np.ascontiguousarray(T0.as_array(), dtype=np.float64)
    return a.tobytes() == b.tobytes()


def shard_range(n, rank, world):
    """contiguous range [lo, hi) of rank `rank`: sizes differ by at most one (round-1 sharding; still
    what the brute-force engine's bench uses)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def block_shard(n_total, rank, world):
    """(first block, end block, blocks, n_local) of rank `rank` (icp_shard_geometry)."""
    b0, b1, bl, nl = C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
    _lib.check(lib().icp_shard_geometry(n_total, rank, world, C.byref(b0), C.byref(b1), C.byref(bl), C.byref(nl)),
               "icp_shard_geometry")
    return b0.value, b1.value, bl.value, nl.value


def local_indices(n_total, rank, world, threads=512):
    """global indices of rank `rank`'s points, in its local (fold) order: chunk `it` of the local arrays
    is the part of the tree's row `it` that its blocks cover."""
    b0, b1, blocks, n_local = block_shard(n_total, rank, world)
    G = blocks * threads
    parts = [np.arange(base + b0 * threads, min(n_total, base + b1 * threads), dtype=np.int64)
             for base in range(0, max(n_total, 1), G) if min(n_total, base + b1 * threads) > base + b0 * threads]
    out = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)
    assert len(out) == n_local
    return out


# ------------------------------------------------------------------------------------ comms ----
class TorchComm:
    """One local rank per process; collectives through torch.distributed (nccl = RCCL, or gloo).  Create the process
    group with a timeout (bench.py: 240 s): the sharded driver keeps the ranks in the same collectives by
    construction (BlockShardedIcp._evaluate), but a rank that dies can only be noticed by its peers that way."""

    def __init__(self, rank, world, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.local_ranks = [rank]

    def sum_(self, bufs):
        import torch.distributed as dist

        if self.world > 1:
            dist.all_reduce(bufs[0], op=dist.ReduceOp.SUM, group=self.group)

    def gather(self, sends, recvs):
        """recvs[0] (world x len) <- every rank's sends[0] (len), rank order"""
        import torch.distributed as dist

        if self.world > 1:
            dist.all_gather_into_tensor(recvs[0].view(-1), sends[0].view(-1), group=self.group)
        else:
            recvs[0].view(-1).copy_(sends[0].view(-1))


    def all_gather_object(self, objs):
        """every rank's objs[0], rank order"""
        import torch.distributed as dist

        if self.world == 1:
            return [objs[0]]
        out = [None] * self.world
        dist.all_gather_object(out, objs[0], group=self.group)
        return out

    def barrier(self):
        import torch.distributed as dist

        if self.world > 1:
            dist.barrier(group=self.group)

    def all_ok(self, ok, like=None):
        """True iff `ok` on EVERY rank (one small all_reduce(MIN)): how the ranks agree on the outcome of an operation
        that went through the mapped inboxes, whose bounded waits can end differently on different ranks"""
        import torch
        import torch.distributed as dist

        if self.world == 1:
            return bool(ok)
        # (RCCL reduces device tensors; gloo -- the CPU tests, ranks sharing one GPU -- host tensors)
        dev = (like.device if like is not None else "cuda") if dist.get_backend(self.group) == "nccl" else "cpu"
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()))


class LocalComm:
    """All ranks live in this process (virtual ranks): the exchanges are plain copies."""

    def __init__(self, world):
        self.world = world
        self.local_ranks = list(range(world))

    def all_gather_object(self, objs):
        return list(objs)

    def barrier(self):
        pass

    def all_ok(self, ok, like=None):
        return bool(ok)

    def sum_(self, bufs):
        total = bufs[0].clone()
        for b in bufs[1:]:
            total += b.to(total.device)
        for b in bufs:
            b.copy_(total)

    def gather(self, sends, recvs):
        for rv in recvs:
            flat = rv.view(self.world, -1)
            for r, s in enumerate(sends):
                flat[r].copy_(s.view(-1))


# ----------------------------------------------------------------------------------- stages ----
class HipStages:
    """The device stages of the C ABI on torch CUDA tensors (include/icp_mi355x.h, sections 4 and 5)."""

    def __init__(self, icp):
        import torch

        self.icp = icp
        self.torch = torch
        icp.set_stream(torch.cuda.current_stream().cuda_stream)
        self.hist_words = int(lib().icp_shard_histogram_words())

    def exch_bytes(self, world):
        """what a rank hands to the others between compact and finish: its candidates + its block sums"""
        return int(lib().icp_shard_exchange_bytes(world))

    def empty(self, nbytes, like):
        return self.torch.empty(nbytes, dtype=self.torch.uint8, device=like.device)

    def empty_points(self, n, cols, like):
        return self.torch.empty((n, cols), dtype=self.torch.float64, device=like.device)

    def empty_index(self, n, like):
        return self.torch.empty(n, dtype=self.torch.int32, device=like.device)

    # -- round-1 stage calls (replicated inner loop) --
    def prepare(self, src_shard, T, presorted=False):
        """the search snapshot of a call; presorted: the shard is a slice of the fold order (take_source after sort_source),
        whose order the snapshot keeps"""
        if presorted:
            _lib.check(lib().icp_shard_prepare_source_device(self.icp._h, C.c_void_p(src_shard.data_ptr()), src_shard.shape[0],
                                                             C.byref(T.pose)), "icp_shard_prepare_source_device")
        else:
            self.icp.prepare_source_device(src_shard, T)

    def correspond(self, src_shard, T, a_out, b_out, idx_out=None):
        self.icp.correspond_device(src_shard, T, a_out, b_out, idx_out)

    def correspond_idx(self, src_shard, T, idx_out):
        self.icp.correspond_device(src_shard, T, None, None, idx_out)

    def materialize(self, src_full, T, idx_full, a_out, b_out):
        self.icp.materialize_pairs_device(src_full, T, idx_full, a_out, b_out)

    def estimate_transform(self, a_full, b_full):
        return self.icp.estimate_transform_device(a_full, b_full)

    def sort_take(self, src_full, T, n_total, rank, world, n_local):
        """(this rank's points in fold order, permutation of the whole cloud): icp_shard_sort_take_device -- the sort of
        sort_source without the sorted copy of the whole cloud"""
        loc = self.torch.empty((max(n_local, 1), src_full.shape[1]), dtype=self.torch.float64, device=src_full.device)[:n_local]
        perm = self.torch.empty(max(n_total, 1), dtype=self.torch.int32, device=src_full.device)
        _lib.check(lib().icp_shard_sort_take_device(self.icp._h, C.c_void_p(src_full.data_ptr()), n_total, C.byref(T.pose), rank, world,
                                                    C.c_void_p(loc.data_ptr()), C.c_void_p(perm.data_ptr())), "icp_shard_sort_take_device")
        return loc, perm[:n_total]

    def sort_source(self, src_full, T):
        """(sorted cloud, permutation): the fold order of a one-GPU estimate call that starts at T
        (icp_sort_source_device); every rank computes the same one from the same inputs"""
        return self.icp.sort_source_device(src_full, T)

    # -- block-sharded evaluation --
    def take(self, full, local, n_total, rank, world):
        _lib.check(lib().icp_shard_take_device(self.icp._h, C.c_void_p(full.data_ptr()), C.c_void_p(local.data_ptr()),
                                               n_total, rank, world, full.element_size() * (full.shape[1] if full.dim() > 1 else 1)),
                   "icp_shard_take_device")

    def put(self, local, full, n_total, rank, world):
        _lib.check(lib().icp_shard_put_device(self.icp._h, C.c_void_p(local.data_ptr()), C.c_void_p(full.data_ptr()),
                                              n_total, rank, world, full.element_size() * (full.shape[1] if full.dim() > 1 else 1)),
                   "icp_shard_put_device")

    def eval_hist(self, a, b, n_total, rank, world, T, kind, refined=False):
        ptr = C.c_void_p()
        rc = lib().icp_shard_eval_hist_device(self.icp._h, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), n_total,
                                              rank, world, C.byref(T.pose), kind, int(refined), C.byref(ptr))
        if not ptr.value:
            return rc, None
        # a tensor view of the handle's histogram buffer (+ its four status words), for the collective (no copy);
        # valid whatever the answer: every rank takes part in the sum
        hist = self._wrap(ptr.value, self.hist_words, a.device)
        return rc, hist

    def eval_status(self, from_device):
        """(ranks that answered OK, RETRY_REPLICATED, NONE, anything else) in the hist stage of the evaluation in flight"""
        out = (C.c_uint32 * 4)()
        _lib.check(lib().icp_shard_eval_status(self.icp._h, out, int(from_device)), "icp_shard_eval_status")
        return tuple(int(x) for x in out)

    def eval_abort(self):
        _lib.check(lib().icp_shard_eval_abort_device(self.icp._h), "icp_shard_eval_abort_device")

    def _wrap(self, ptr, words, device):
        key = (ptr, words)
        cache = self.__dict__.setdefault("_views", {})
        if key not in cache:
            torch = self.torch

            class _Raw:  # __cuda_array_interface__ of a foreign device allocation
                pass

            raw = _Raw()
            raw.__cuda_array_interface__ = {"shape": (words,), "typestr": "<i4", "data": (ptr, False), "version": 2}
            cache[key] = torch.as_tensor(raw, device=device)
        return cache[key]

    def eval_compact(self, exch_out):
        return lib().icp_shard_eval_compact_device(self.icp._h, C.c_void_p(exch_out.data_ptr()))

    # -- the inner loop as one launch per rank (include/icp_mi355x.h section 5b) --
    def loop_inbox(self, kind=0):
        """device pointer of this rank's inbox (allocated on first use); kind: 0 device memory, 1 fine-grained device
        memory, 2 pinned host memory in a shared-memory object (include/icp_mi355x.h: ICP_INBOX_*)"""
        ptr = C.c_void_p()
        _lib.check(lib().icp_loop_inbox(self.icp._h, int(kind), C.byref(ptr)), "icp_loop_inbox")
        return int(ptr.value)

    def loop_ipc_handle(self):
        buf = C.create_string_buffer(64)
        _lib.check(lib().icp_loop_inbox_ipc_handle(self.icp._h, buf), "icp_loop_inbox_ipc_handle")
        return bytes(buf.raw)

    def loop_ipc_open(self, handle, device):
        ptr = C.c_void_p()
        _lib.check(lib().icp_loop_ipc_open(int(device), C.create_string_buffer(handle, 64), C.byref(ptr)), "icp_loop_ipc_open")
        return int(ptr.value)

    def loop_ipc_close(self, ptr):
        lib().icp_loop_ipc_close(C.c_void_p(int(ptr)))

    def loop_shm_name(self):
        buf = C.create_string_buffer(64)
        _lib.check(lib().icp_loop_inbox_shm_name(self.icp._h, buf), "icp_loop_inbox_shm_name")
        return bytes(buf.value)

    def loop_shm_unlink(self):
        lib().icp_loop_inbox_shm_unlink(self.icp._h)

    def loop_shm_open(self, name, device):
        ptr = C.c_void_p()
        _lib.check(lib().icp_loop_shm_open(int(device), C.c_char_p(name), C.byref(ptr)), "icp_loop_shm_open")
        return int(ptr.value)

    def loop_shm_close(self, ptr):
        lib().icp_loop_shm_close(C.c_void_p(int(ptr)))

    def loop_probe(self, rounds=8):
        ok = C.c_int(0)
        _lib.check(lib().icp_loop_transport_probe(self.icp._h, int(rounds), C.byref(ok)), "icp_loop_transport_probe")
        return bool(ok.value)

    def reset_predictions(self):
        _lib.check(lib().icp_reset_window_predictions(self.icp._h), "icp_reset_window_predictions")

    def device_id(self):
        """what tells two devices of one node apart (ranks that share a device share its L2)"""
        dev = self.torch.cuda.current_device()
        buf = C.create_string_buffer(64)
        if lib().icp_device_pci_bus_id(int(dev), buf) == _lib.OK and buf.value:
            return buf.value.decode(), dev  # (the same for every process that sees this GPU, whatever its ordinal there)
        props = self.torch.cuda.get_device_properties(dev)
        return str(getattr(props, "uuid", None) or getattr(props, "pci_bus_id", None) or dev), dev

    def loop_connect(self, rank, world, inbox_ptrs):
        arr = (C.c_void_p * world)(*[C.c_void_p(int(p)) for p in inbox_ptrs])
        _lib.check(lib().icp_shard_loop_connect(self.icp._h, rank, world, arr), "icp_shard_loop_connect")

    def loop_launch(self, a, b, n_total, launch_no, eval_base, it0, applied, Ti, prev_error, first_kind=0, second_kind=1):
        return lib().icp_shard_loop_launch_device(self.icp._h, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), n_total,
                                                  launch_no, eval_base, it0, applied, C.byref(Ti.pose), prev_error,
                                                  first_kind, second_kind)

    def loop_wait(self):
        """(rc, Ti, prev_error, applied, it, finished, evaluations served)"""
        Ti = Transform()
        pe, ap, it, fin, ev = C.c_double(0.0), C.c_uint32(0), C.c_int(0), C.c_int(0), C.c_uint32(0)
        rc = lib().icp_shard_loop_wait(self.icp._h, C.byref(Ti.pose), C.byref(pe), C.byref(ap), C.byref(it), C.byref(fin),
                                       C.byref(ev))
        return rc, Ti, pe.value, ap.value, it.value, bool(fin.value), ev.value

    def pipe_run(self, src_local, n_total, rank, world, T, it, max_iter, inner, idx_out):
        """icp_shard_pipe_run_device (include/icp_mi355x.h section 5c): this rank's iterations of the pipelined sharded
        registration from outer iteration `it` on.  inner: uint32 numpy array of max_iter entries (filled for the
        iterations served); idx_out: int32 device tensor for the last search's correspondences.  Returns (T, it, why)."""
        To = Transform()
        To.pose = type(T.pose).from_buffer_copy(T.pose)
        itc, why = C.c_size_t(int(it)), C.c_int(1)
        _lib.check(lib().icp_shard_pipe_run_device(self.icp._h, C.c_void_p(src_local.data_ptr()), src_local.shape[0], n_total, rank,
                                                   world, C.byref(To.pose), C.byref(itc), max_iter,
                                                   C.c_void_p(inner.ctypes.data), C.c_void_p(idx_out.data_ptr()), C.byref(why)),
                   "icp_shard_pipe_run_device")
        return To, int(itc.value), int(why.value)

    def pipe_counters(self):
        out = (C.c_uint64 * 4)()
        _lib.check(lib().icp_pipe_counters(self.icp._h, out), "icp_pipe_counters")
        return tuple(int(x) for x in out)

    def eval_finish(self, exch_all):
        delta = np.zeros(3)
        err = C.c_double(0.0)
        rc = lib().icp_shard_eval_finish_device(self.icp._h, C.c_void_p(exch_all.data_ptr()),
                                                delta.ctypes.data_as(C.POINTER(C.c_double)), C.byref(err))
        return rc, delta, err.value

    def gn_step(self, a_full, b_full, T, kind):
        delta = np.zeros(3)
        err = C.c_double(0.0)
        rc = lib().icp_weighted_gn_step_device(self.icp._h, C.c_void_p(a_full.data_ptr()), C.c_void_p(b_full.data_ptr()),
                                               a_full.shape[0], C.byref(T.pose), kind,
                                               delta.ctypes.data_as(C.POINTER(C.c_double)), C.byref(err))
        return rc, delta, err.value


# ------------------------------------------------------------------------ block-sharded driver ----
class _GaveUp(Exception):
    """some rank's wait for a peer ran out inside a launch that exchanges through the mapped inboxes (agreed on by all
    ranks: BlockShardedIcp.estimate starts the call again through the stage calls + collectives)"""


class _Rank:
    def __init__(self, rank, stages):
        self.rank, self.stages = rank, stages
        self.bufs = None


class BlockShardedIcp:
    """Icp{2,3}d::estimate (src/lib.rs:105-130, 148-173) with the source cloud sharded by reduction-tree
    block.  `stages_by_rank`: {rank: stages} for the ranks that live in this process (one for
    torch.distributed; all of them for virtual ranks); `comm`: TorchComm or LocalComm."""

    def __init__(self, stages_by_rank, n_total, world, comm, mul=None):
        self.n, self.world, self.comm = n_total, world, comm
        self.ranks = [_Rank(r, stages_by_rank[r]) for r in comm.local_ranks]
        self._mul = mul or (lambda a, b: a * b)
        self._new = lambda d: Transform(d)
        self.geom = {r: block_shard(n_total, r, world) for r in range(world)}
        self.n_local_max = max(g[3] for g in self.geom.values())
        self.counters = {"sharded": 0, "replicated": 0}

    TRANSPORTS = {"device_ipc": 0, "fine_ipc": 1, "host_shm": 2}

    def connect_loop(self, fine_grained=False, transport="auto", probe=True):
        """Collective over the ranks: map every rank's inbox on every rank, so that an inner loop is ONE launch per rank
        whose workgroups exchange histograms, candidates and block sums through memory (include/icp_mi355x.h section 5b;
        gn_loop.hip).  Ranks of one process hand each other plain pointers (fine_grained: they sit on DISTINCT devices
        and use peer access).  Across processes a TRANSPORT carries the inboxes (one small all_gather_object at connect
        time -- nothing of the per-iteration path goes through torch.distributed, only evaluations a launch hands back):
          device_ipc  ordinary device memory through hipIpc -- for processes that share ONE device (its L2 keeps them
                      coherent); between distinct devices such memory is coherent at kernel boundaries only
          fine_ipc    fine-grained device memory through hipIpc (where the runtime exports it)
          host_shm    pinned host memory in a POSIX shared-memory object, registered by every process
        "auto": device_ipc when every rank reports the same device, else fine_ipc, then host_shm -- the first whose
        ping-pong probe (icp_loop_transport_probe: every rank at once, bounded waits) passes on EVERY rank.  Returns the
        transport's name, or None: no transport passed, the stage calls + collectives serve (self._loop stays unset)."""
        import os

        self._loop = None
        self.loop_transport = None
        self._loop_opened = []
        self.counters.update(loop_launches=0, loop_served=0, loop_handbacks=0, loop_gave_up=0)
        if isinstance(self.comm, LocalComm):
            mine = {rk.rank: rk.stages.loop_inbox(1 if fine_grained else 0) for rk in self.ranks}
            for rk in self.ranks:
                rk.stages.loop_connect(rk.rank, self.world, [mine[r] for r in range(self.world)])
            self._loop = dict(launch=0, evals=0)
            self.loop_transport = "pointers (one process" + (", peer access)" if fine_grained else ")")
            return self.loop_transport
        rk = self.ranks[0]
        st = rk.stages
        dev_id, dev = st.device_id() if hasattr(st, "device_id") else ("0", 0)
        ids = self.comm.all_gather_object([(os.uname().nodename, dev_id)])
        if len({h for h, _ in ids}) > 1:
            return None  # (more than one node: no memory to share)
        same_device = len({d for _, d in ids}) == 1
        # (auto: the remaining transports stay as fallbacks -- ranks that see one GPU each through HIP_VISIBLE_DEVICES may
        # all report "device 0" while sitting on different devices, where device_ipc fails its probe: ADVICE r5)
        order = [transport] if transport != "auto" else (["device_ipc", "fine_ipc", "host_shm"] if same_device
                                                         else ["fine_ipc", "host_shm"])
        for name in order:
            kind = self.TRANSPORTS[name]
            ok, err, opened, ptrs = 1, "", [], []
            try:
                raw = st.loop_inbox(kind)
                token = st.loop_shm_name() if kind == 2 else st.loop_ipc_handle()
            except Exception as e:  # noqa: BLE001  (e.g. the runtime does not export a fine-grained allocation)
                ok, err, raw, token = 0, repr(e), 0, b""
            infos = self.comm.all_gather_object([(os.getpid(), ok, token, raw)])
            ok = min(i[1] for i in infos)
            if ok:
                try:
                    for r, (pid, _, tok, rawp) in enumerate(infos):
                        if r == rk.rank:
                            ptrs.append(raw)
                        elif pid == os.getpid():
                            ptrs.append(rawp)
                        else:
                            ptr = st.loop_shm_open(tok, dev) if kind == 2 else st.loop_ipc_open(tok, dev)
                            opened.append((kind, ptr))
                            ptrs.append(ptr)
                    st.loop_connect(rk.rank, self.world, ptrs)
                except Exception as e:  # noqa: BLE001
                    ok, err = 0, repr(e)
            ok = min(self.comm.all_gather_object([ok]))
            self.comm.barrier()  # (a connect empties the own inbox: nobody may write before everybody has connected)
            if kind == 2:
                st.loop_shm_unlink()  # every peer has opened it: the name can go, the mappings live on
            if ok and probe:
                try:
                    ok = 1 if st.loop_probe() else 0
                except Exception as e:  # noqa: BLE001
                    ok, err = 0, repr(e)
                ok = min(self.comm.all_gather_object([ok]))
            if ok:
                self._loop = dict(launch=0, evals=0)
                self._loop_opened = opened
                self.loop_transport = name
                return name
            self._close_opened(opened)
            self.comm.barrier()
        return None

    def _close_opened(self, opened):
        st = self.ranks[0].stages
        for kind, ptr in opened:
            (st.loop_shm_close if kind == 2 else st.loop_ipc_close)(ptr)

    def disconnect_loop(self):
        """Collective: give the peers' inboxes back (the mappings connect_loop opened), after a barrier -- nobody may
        still be launching into them -- and before any rank frees or pools its handle."""
        self._loop = None
        if not isinstance(self.comm, LocalComm):
            self.comm.barrier()
            self._close_opened(getattr(self, "_loop_opened", []))
            self._loop_opened = []
            self.comm.barrier()

    def _loop_run(self, Ti, prev_error, applied, it, first_kind=0, second_kind=1):
        """the inner loop from evaluation `it` on as one launch per local rank; None: nothing was launched, or the
        launch gave up (the loop's state is then the one it was started with, and the stage calls serve from here on)"""
        import os

        L = self._loop
        launch_no = L["launch"] + 1
        rcs = []
        # TEST HOOK (tests/test_gpu_ipc.py): ICP_DIST_TEST_WITHHOLD="<rank>:<launch>" -- that rank does not launch that
        # inner loop, i.e. withholds every flag its peers wait for: their launches run into their bounded waits (3 s), raise
        # abort in every inbox and report it; this rank reports the same; every rank agrees (comm.all_ok below) and the
        # call starts again through the stage calls + collectives.  Never set outside the test.
        hook = os.environ.get("ICP_DIST_TEST_WITHHOLD")
        if hook and not isinstance(self.comm, LocalComm) and self.world > 1:
            r_h, l_h = (int(x) for x in hook.split(":"))
            if launch_no == l_h:
                if self.ranks[0].rank == r_h:
                    if not self.comm.all_ok(False, like=self.ranks[0].bufs["a"]):
                        raise _GaveUp()
        for rk in self.ranks:
            nl = self.geom[rk.rank][3]
            rcs.append(rk.stages.loop_launch(rk.bufs["a"][:nl], rk.bufs["b"][:nl], self.n, launch_no, L["evals"], it, applied, Ti,
                                             prev_error, first_kind, second_kind))
        if all(rc == _lib.RETRY_SHARDED for rc in rcs):
            return None
        for rc in rcs:
            _lib.check(rc, "icp_shard_loop_launch_device")
        L["launch"] = launch_no
        outs = [rk.stages.loop_wait() for rk in self.ranks]
        gave_up = any(o[0] == _lib.HIP_ERROR for o in outs)
        if not isinstance(self.comm, LocalComm) and self.world > 1:
            #  the bounded waits of a launch can end differently on different ranks -- one finishes its last wait
            # while a peer's runs out -- so the hosts AGREE on the outcome before anybody uses it: one small all_reduce
            if not self.comm.all_ok(not gave_up, like=self.ranks[0].bufs["a"]):
                raise _GaveUp()
        if gave_up:
            # A launch gave up waiting for a peer.  The rank whose wait ran out raised the abort word in EVERY inbox, so
            # every rank's launch ended the same way and every host is here (no collective needed to agree).  Nothing
            # of the launch is used; the prediction histories may have diverged inside it and are dropped; the stage
            # calls + collectives serve this evaluation and all later ones (same bits either way).
            for rk in self.ranks:
                if hasattr(rk.stages, "reset_predictions"):
                    rk.stages.reset_predictions()
            self._loop = None
            self.counters["loop_gave_up"] += 1
            return None
        for o in outs:
            _lib.check(o[0], "icp_shard_loop_wait", allow=(_lib.NAN_INPUT,))
        o = outs[0]
        assert all(x[0] == o[0] and x[1].as_array().tobytes() == o[1].as_array().tobytes() and x[2:] == o[2:] for x in outs)
        L["evals"] += o[6]
        self.counters["loop_launches"] += 1
        self.counters["loop_served"] += o[6]
        self.counters["sharded"] += o[6]
        if not o[5]:
            self.counters["loop_handbacks"] += 1
        return o

    def set_pose_algebra(self, new, mul):
        """(tests) Transform::new / Mul implementations; default: the C ABI's"""
        self._new, self._mul = new, mul

    def take_source(self, src_full_by_rank):
        """compact each local rank's points out of a full source cloud -> {rank: local cloud}"""
        out = {}
        for rk in self.ranks:
            full = src_full_by_rank[rk.rank] if isinstance(src_full_by_rank, dict) else src_full_by_rank
            loc = rk.stages.empty_points(self.geom[rk.rank][3], full.shape[1], full)
            rk.stages.take(full, loc, self.n, rk.rank, self.world)
            out[rk.rank] = loc
        return out

    def sort_source(self, src_full_by_rank, T):
        """The one-GPU path folds its sums over the source cloud in FOLD ORDER (include/icp_mi355x.h,
        icp_last_fold_order: a deterministic sort by target-grid cell under the call's initial pose).  To
        return the same bits, the sharded path shards THAT order: every local rank sorts the full cloud
        (same inputs, same result everywhere, no communication) -> ({rank: sorted cloud}, {rank: permutation});
        stages without a sort (the CPU stand-ins of the tests) keep the caller's order."""
        out, perms = {}, {}
        for rk in self.ranks:
            full = src_full_by_rank[rk.rank] if isinstance(src_full_by_rank, dict) else src_full_by_rank
            if hasattr(rk.stages, "sort_source"):
                out[rk.rank], perms[rk.rank] = rk.stages.sort_source(full, T)
            else:
                out[rk.rank], perms[rk.rank] = full, None
        return out, perms

    def estimate_full(self, src_full_by_rank, initial_transform, max_iter):
        """Icp::estimate from the full source cloud, as one GPU runs it: fold order, shard, iterate.
        Returns (T, inner, {rank: permutation of the fold order or None})."""
        if all(hasattr(rk.stages, "sort_take") for rk in self.ranks):
            # (the fold order of the whole cloud and the rank's points gathered through it: no sorted copy of the whole cloud)
            local, perms = {}, {}
            for rk in self.ranks:
                full = src_full_by_rank[rk.rank] if isinstance(src_full_by_rank, dict) else src_full_by_rank
                local[rk.rank], perms[rk.rank] = rk.stages.sort_take(full, initial_transform, self.n, rk.rank, self.world,
                                                                      self.geom[rk.rank][3])
        else:
            srt, perms = self.sort_source(src_full_by_rank, initial_transform)
            local = self.take_source(srt)
        self._presorted = all(p is not None for p in perms.values())  # (the ranks' slices are runs of the sorted cloud)
        try:
            T, inner = self.estimate(local, initial_transform, max_iter)
        finally:
            self._presorted = False
        return T, inner, perms

    def _buffers(self, rk, like):
        if rk.bufs is None:
            st, W = rk.stages, self.world
            nl = max(self.geom[rk.rank][3], 1)
            rk.bufs = dict(
                a=st.empty_points(nl, 2, like), b=st.empty_points(nl, 2, like), idx=st.empty_index(nl, like),
                exch=st.empty(st.exch_bytes(W), like), exch_all=st.empty(W * st.exch_bytes(W), like),
                # replicated fallback: every rank's pairs, padded to the largest shard, and the full arrays
                pair_send=st.empty_points(self.n_local_max, 4, like),
                pair_recv=st.empty_points(W * self.n_local_max, 4, like),
                a_full=st.empty_points(max(self.n, 1), 2, like), b_full=st.empty_points(max(self.n, 1), 2, like),
                have_full=False)
        return rk.bufs

    # one evaluation of weighted_gauss_newton_update at inner pose T, on every local rank
    def _evaluate(self, T, kind, refined=False):
        rks = self.ranks
        res = [rk.stages.eval_hist(rk.bufs["a"][:self.geom[rk.rank][3]], rk.bufs["b"][:self.geom[rk.rank][3]], self.n,
                                   rk.rank, self.world, T, kind, refined) for rk in rks]
        rcs = {rc for rc, _ in res}
        if all(hasattr(rk.stages, "eval_status") for rk in rks):
            # Every rank takes part in the SAME two exchanges whatever its own hist stage answered; the answers travel
            # as four status counters behind the histograms, so after the sum all ranks know all answers and branch
            # alike.  Each rank decides its answer from state that is replicated by construction (the window
            # prediction comes from global statistics), but a rank-local condition -- a HIP error, a handle with
            # stale state -- would otherwise send it into a different collective than its peers, which hang.
            for rc, h in res:
                if h is None:
                    _lib.check(rc, "icp_shard_eval_hist_device")  # (bad arguments: the same on every rank)
            self.comm.sum_([h for _, h in res])
            ok = [rc == _lib.OK for rc, _ in res]
            for rk, o in zip(rks, ok):
                if o:
                    _lib.check(rk.stages.eval_compact(rk.bufs["exch"]), "icp_shard_eval_compact_device")
                else:
                    rk.bufs["exch"].zero_()
            self.comm.gather([rk.bufs["exch"] for rk in rks], [rk.bufs["exch_all"] for rk in rks])
            outs, sts = [], []
            for rk, o in zip(rks, ok):
                outs.append(rk.stages.eval_finish(rk.bufs["exch_all"]) if o else None)
                sts.append(rk.stages.eval_status(not o))
            st = sts[0]
            assert all(s == st for s in sts)
            if st[0] == self.world:  # every rank evaluated its share
                rc = outs[0][0]
                assert all(o[0] == rc for o in outs)  # every rank folds the same numbers
                if rc == _lib.RETRY_SHARDED:  # the window missed; its counts place one that will not
                    self.counters["refined"] = self.counters.get("refined", 0) + 1
                    return self._evaluate(T, kind, refined=True)
                if rc != _lib.RETRY_REPLICATED:
                    self.counters["sharded"] += 1
                    return outs[0]
            else:
                for rk in rks:
                    rk.stages.eval_abort()
                if st[2] == self.world:
                    return _lib.NONE, None, 0.0
                if st[1] != self.world:  # the ranks disagree (or one failed): the same exception on every rank
                    raise RuntimeError(f"sharded evaluation: the ranks' hist stages answered differently "
                                       f"(OK, RETRY_REPLICATED, NONE, other) = {st}")
        elif rcs == {_lib.OK}:
            self.comm.sum_([h for _, h in res])
            for rk in rks:
                _lib.check(rk.stages.eval_compact(rk.bufs["exch"]), "icp_shard_eval_compact_device")
            self.comm.gather([rk.bufs["exch"] for rk in rks], [rk.bufs["exch_all"] for rk in rks])
            outs = [rk.stages.eval_finish(rk.bufs["exch_all"]) for rk in rks]
            rc = outs[0][0]
            assert all(o[0] == rc for o in outs)  # every rank folds the same numbers
            if rc == _lib.RETRY_SHARDED:  # the window missed; its counts place one that will not
                self.counters["refined"] = self.counters.get("refined", 0) + 1
                return self._evaluate(T, kind, refined=True)
            if rc != _lib.RETRY_REPLICATED:
                self.counters["sharded"] += 1
                return outs[0]
        elif rcs == {_lib.NONE}:
            return _lib.NONE, None, 0.0
        elif rcs != {_lib.RETRY_REPLICATED}:
            _lib.check(max(rcs), "icp_shard_eval_hist_device")
        # replicated: gather the pairs of all ranks into global order, evaluate them on every rank
        self.counters["replicated"] += 1
        if not rks[0].bufs["have_full"]:
            for rk in rks:
                nl = self.geom[rk.rank][3]
                rk.bufs["pair_send"][:nl, 0:2] = rk.bufs["a"][:nl]
                rk.bufs["pair_send"][:nl, 2:4] = rk.bufs["b"][:nl]
            self.comm.gather([rk.bufs["pair_send"] for rk in rks], [rk.bufs["pair_recv"] for rk in rks])
            for rk in rks:
                recv = rk.bufs["pair_recv"].view(self.world, self.n_local_max, 4)
                for q in range(self.world):
                    nq = self.geom[q][3]
                    if nq:
                        rk.stages.put(recv[q, :nq, 0:2].contiguous(), rk.bufs["a_full"], self.n, q, self.world)
                        rk.stages.put(recv[q, :nq, 2:4].contiguous(), rk.bufs["b_full"], self.n, q, self.world)
                rk.bufs["have_full"] = True
        outs = [rk.stages.gn_step(rk.bufs["a_full"][:self.n], rk.bufs["b_full"][:self.n], T, kind) for rk in rks]
        assert all(o[0] == outs[0][0] for o in outs)
        return outs[0]

    def step(self, src_local, T):
        """one outer iteration (src/lib.rs:113-127 / 156-170); src_local: {rank: local cloud}.
        Returns (dT * T, inner iterations applied)."""
        for rk in self.ranks:
            s = src_local[rk.rank]
            bf = self._buffers(rk, s)
            nl = self.geom[rk.rank][3]
            bf["have_full"] = False
            if nl:
                rk.stages.correspond(s, T, bf["a"][:nl], bf["b"][:nl], bf["idx"][:nl])
        # estimate_transform, src/lib.rs:59-84
        Ti = Transform()
        applied = 0
        # the kinds of this inner loop's first two evaluations (csrc/common.hpp: Workspace::win_kind): a call's first outer
        # iteration is predicted from the previous call's first iteration (3, 4), every other from the one before it (0, 1)
        ka, kb = (3, 4) if getattr(self, "_outer_it", 1) == 0 else (0, 1)
        if self.n >= 2:
            prev_error = float(np.finfo(np.float64).max)
            it = 0
            while it < INNER_MAX_ITER:
                if getattr(self, "_loop", None) is not None:
                    o = self._loop_run(Ti, prev_error, applied, it, ka, kb)
                    if o is not None:
                        rc, Ti, prev_error, applied, it, finished, _ = o
                        _lib.check(rc, "icp_shard_loop_wait")
                        if finished or it >= INNER_MAX_ITER:
                            break
                rc, delta, err = self._evaluate(Ti, ka if it == 0 else (kb if it == 1 else 2))
                if rc == _lib.NONE:
                    break
                _lib.check(rc, "sharded evaluation")
                if (delta[0] * delta[0] + delta[1] * delta[1]) + delta[2] * delta[2] < DELTA_NORM_THRESHOLD:
                    break
                if err > prev_error:
                    break
                prev_error = err
                Ti = self._mul(self._new(delta), Ti)
                applied += 1
                it += 1
        return self._mul(Ti, T), applied

    def _pipe_usable(self):
        """the pipelined evaluation (csrc/pipe.hip) serves ONE local rank per process (ranks of one process on one stream
        would wait for kernels that are not enqueued yet: icp_create_multi fuses their launches instead) over connected
        inboxes, every rank with at most 256 tree blocks (N ranks: N x 2^20 points); ICP_DIST_NO_PIPE=1 switches it off"""
        import os

        return (getattr(self, "_loop", None) is not None and len(self.ranks) == 1 and
                hasattr(self.ranks[0].stages, "pipe_run") and os.environ.get("ICP_DIST_NO_PIPE") != "1" and
                all(g[3] > 0 and g[1] - g[0] <= 256 for g in self.geom.values()))

    def estimate(self, src_local, initial_transform, max_iter):
        try:
            return self._estimate(src_local, initial_transform, max_iter)
        except _GaveUp:
            # Agreed on by every rank (comm.all_ok): a launch that exchanges through the inboxes gave up somewhere.  Nothing
            # of this call is kept -- the ranks may have got one iteration apart -- the inbox paths are off for the rest of
            # the connection, the prediction histories are dropped, and the call starts again through the stage calls +
            # collectives, which every rank enters at the same point (same bits either way).
            for rk in self.ranks:
                if hasattr(rk.stages, "reset_predictions"):
                    rk.stages.reset_predictions()
            self._loop = None
            self.counters["loop_gave_up"] = self.counters.get("loop_gave_up", 0) + 1
            return self._estimate(src_local, initial_transform, max_iter)

    def _estimate(self, src_local, initial_transform, max_iter):
        T = initial_transform
        inner = np.zeros(max_iter, dtype=np.uint32)
        if max_iter > 0:
            for rk in self.ranks:
                if hasattr(rk.stages, "prepare") and self.geom[rk.rank][3]:
                    if getattr(self, "_presorted", False) and isinstance(rk.stages, HipStages):
                        rk.stages.prepare(src_local[rk.rank], T, presorted=True)
                    else:
                        rk.stages.prepare(src_local[rk.rank], T)
        # (the bet needs "the inner loop applied exactly one update last time": a call's first iteration goes by what the
        # previous call's first iteration did -- the next frame, or the same cloud again, usually starts like the last one)
        it, prev_k, skip = 0, getattr(self, "_first_k", None), 0
        while it < max_iter:
            # Round 6: once an inner loop has applied exactly one update, the rank runs the one-GPU pipeline with finishing
            # workgroups that meet its peers' (HipStages.pipe_run) until something else happens; this loop then serves
            # that iteration and may come back.
            if prev_k == 1 and skip == 0 and self._pipe_usable():
                rk = self.ranks[0]
                bf = self._buffers(rk, src_local[rk.rank])
                nl = self.geom[rk.rank][3]
                T2, it2, why = rk.stages.pipe_run(src_local[rk.rank], self.n, rk.rank, self.world, T, it, max_iter, inner,
                                                  bf["idx"][:nl])
                if not self.comm.all_ok(why != 5, like=bf["a"]):
                    raise _GaveUp()
                self.counters["pipe_iterations"] = self.counters.get("pipe_iterations", 0) + (it2 - it)
                self.counters["sharded"] += 2 * (it2 - it)
                if it2 == it:
                    skip = 2  # (handed back at once: a few iterations through this loop before the next try)
                elif it == 0:
                    self._first_k = 1
                T, it = T2, it2
                if it >= max_iter:
                    break
            elif skip > 0:
                skip -= 1
            self._outer_it = it
            T_next, k = self.step(src_local, T)
            if it == 0:
                self._first_k = k
            inner[it] = k
            prev_k = k
            it += 1
            if k == 0 and _same_bits(T_next, T):  # a fixed point of the loop: the iterations after it repeat it
                break  # (every step leaves its indices: the last need not run; the remaining inner counts are 0)
            T = T_next
        return T, inner

    def last_indices(self):
        """{rank: correspondence indices of its local points in the last outer iteration}"""
        return {rk.rank: rk.bufs["idx"][:self.geom[rk.rank][3]] for rk in self.ranks}


# ------------------------------------------------- round-1 driver: replicated inner loop ----------
class ShardedIcp:
    """Contiguous source shards, pairs (or indices) all-gathered once per outer iteration, the whole
    inner loop replicated on every rank.  Kept for the brute-force engine, whose search is 99.8 % of the
    step (the search is all that needs to shard there), and as the reference the block-sharded driver is
    compared with."""

    def __init__(self, stages, n_total, rank=0, world=1, group=None, mul=None, src_full=None):
        self.stages = stages
        self.n = n_total
        self.src_full = src_full  # optional: the whole source cloud on every rank -> index gather
        self.rank, self.world, self.group = rank, world, group
        self.lo, self.hi = shard_range(n_total, rank, world)
        self.max_shard = shard_range(n_total, 0, world)[1]
        self._mul = mul or (lambda a, b: a * b)
        self._bufs = None

    def _buffers(self, like):
        import torch

        if self._bufs is None:
            kw = dict(dtype=torch.float64, device=like.device)
            W, ms = self.world, self.max_shard
            even = self.n % W == 0
            gathered = torch.empty((2, W * ms, 2), **kw)  # [a|b] x rank-major padded shards
            local = torch.empty((2, ms, 2), **kw)
            full = gathered if even else torch.empty((2, self.n, 2), **kw)
            self._bufs = (local, gathered, full, even)
        return self._bufs

    def _step_idx(self, src_shard, T):
        """index-gather variant: 4 B/point on the wire, pairs rebuilt locally."""
        import torch
        import torch.distributed as dist

        if getattr(self, "_ibufs", None) is None:
            W, ms = self.world, self.max_shard
            kw = dict(device=src_shard.device)
            self._ibufs = (torch.empty(ms, dtype=torch.int32, **kw), torch.empty(W * ms, dtype=torch.int32, **kw),
                           torch.empty(self.n, dtype=torch.int32, **kw),
                           torch.empty((2, self.n, 2), dtype=torch.float64, **kw))
        loc, gat, idx_full, ab = self._ibufs
        ns = self.hi - self.lo
        self.stages.correspond_idx(src_shard, T, loc[:ns])
        dist.all_gather_into_tensor(gat, loc, group=self.group)
        if self.n % self.world == 0:
            idx_full = gat
        else:
            ms = self.max_shard
            for r in range(self.world):
                lo, hi = shard_range(self.n, r, self.world)
                idx_full[lo:hi] = gat[r * ms: r * ms + (hi - lo)]
        self.stages.materialize(self.src_full, T, idx_full[: self.n], ab[0], ab[1])
        dT, inner = self.stages.estimate_transform(ab[0], ab[1])
        return self._mul(dT, T), inner

    def step(self, src_shard, T):
        """one outer iteration; returns (dT * T, inner_iters)."""
        import torch.distributed as dist

        if self.world > 1 and self.src_full is not None and hasattr(self.stages, "materialize"):
            return self._step_idx(src_shard, T)
        local, gathered, full, even = self._buffers(src_shard)
        ns = self.hi - self.lo
        self.stages.correspond(src_shard, T, local[0, :ns], local[1, :ns])
        if self.world > 1:
            # a and b travel in one collective each so that the receive layout is rank-major
            dist.all_gather_into_tensor(gathered[0], local[0], group=self.group)
            dist.all_gather_into_tensor(gathered[1], local[1], group=self.group)
            if not even:
                ms = self.max_shard
                for r in range(self.world):
                    lo, hi = shard_range(self.n, r, self.world)
                    full[:, lo:hi] = gathered[:, r * ms: r * ms + (hi - lo)]
        else:
            full = local
        dT, inner = self.stages.estimate_transform(full[0, : self.n], full[1, : self.n])
        return self._mul(dT, T), inner

    def estimate(self, src_shard, initial_transform, max_iter):
        """Icp{2,3}d::estimate over the sharded source (src/lib.rs:105-130, 148-173)."""
        T = initial_transform
        inner = []
        if max_iter > 0 and hasattr(self.stages, "prepare"):
            self.stages.prepare(src_shard, T)  # once per estimate call, like Icp::estimate itself
        for _ in range(max_iter):
            T_next, k = self.step(src_shard, T)
            inner.append(k)
            if k == 0 and _same_bits(T_next, T):  # a fixed point of the loop: the iterations after it repeat it
                inner.extend([0] * (max_iter - len(inner)))
                break
            T = T_next
        return T, np.array(inner, dtype=np.uint32)
