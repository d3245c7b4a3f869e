// The PIPELINED sharded registration (round 6): a rank's outer iteration in the steady state of a registration is the
// one-GPU pipeline of api.hip: icp_estimate_device (src/lib.rs:105-130, 148-173 with the inner loop :59-84 applying one
// update) -- search -> both evaluations' first launches -> their finishing workgroups -> the search that was enqueued
// behind them -- with the ranks meeting INSIDE the finishing workgroups (gn_win.hip: k_win_pick_shard) instead of in
// collectives or in a persistent launch.  Per outer iteration a rank enqueues three launches and waits once.
//
//   steady state   the previous inner loop applied exactly ONE update, and the handle has window predictions for both
//                  kinds of evaluation (0: first evaluation on new correspondences, 1: the evaluation after the update).
//   the bet        after the first evaluation E1(k) the pose of iteration k + 1 is known if the deciding evaluation
//                  E2(k) ends the loop: T(k+1) = Exp(delta_1) T(k).  The search for it is enqueued at once (or is in
//                  flight already: the run-ahead search read the same pose from device memory, where E1(k)'s finishing
//                  workgroup left it), and E1(k+1) rides in ONE launch with E2(k).
//   handing back   anything else -- a window that missed, an inner loop of no or several updates, a NaN -- ends the
//                  pipeline at the START of that outer iteration: the caller's own loop (stage calls + collectives, or
//                  the one-launch inner loop) serves it and may come back.  Results are the same bits either way: the
//                  evaluations are the sharded evaluations of shard.hip / gn_loop.hip, folded in the same order.
//
// One function serves both hosts: icp_multi (all ranks in this process, in lockstep from one thread; ranks of one device
// share a stream and a finishing launch) and one process per GPU (a group of one rank; its peers run the same code and
// take the same decisions from the same bits -- icp_shard_pipe_run_device).
#include <cfloat>
#include <cstring>

#include "api_internal.hpp"

using namespace icp;
using namespace icp::api;

namespace icp {

namespace {

inline double norm2(const double d[3]) { return (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]; }

struct PipeWin {
  WinParams P;
  bool own;
  double med[2], sigma[2];  // the prediction the window is centred on
};

// the window of an evaluation of `kind` from the handle's own history of that kind (replicated state: every rank has
// recorded the same statistics); false: no prediction, or its fine windows hold more members than a workgroup can file
bool pipe_window(const icp_handle *h, size_t n_total, int kind, PipeWin *out) {
  const Workspace &w = h->ws;
  if (!Workspace::kind_has_slot(kind) || !w.win_kind[kind].valid) return false;
  if (!window_usable(h, n_total, &out->P, kind, true)) return false;
  out->own = true;
  for (int d = 0; d < 2; ++d) {
    out->med[d] = w.win_kind[kind].med[d];
    out->sigma[d] = w.win_kind[kind].sigma[d];
  }
  return bkt_fits_rank(n_total, out->P);
}

// what an evaluation leaves in the prediction history (api.hip: wgn_step does the same for one handle)
void pipe_record(Workspace &w, int kind, const PipeWin &win, const GnResult &r) {
  bool &wide = w.win_kind[kind].wide;
  if (wide) {
    double shift = 0.;
    for (int d = 0; d < 2; ++d)
      shift = fmax(shift, (fabs(r.median[d] - win.med[d]) + fabs(r.sigma[d] - win.sigma[d])) / win.sigma[d]);
    if (shift < 0.01) wide = false;
  }
  record_statistics(w, kind, true, r);
}

}  // namespace

// Iterations *it_io .. of the registration, as far as the steady state lasts.  *why = 0: through max_iter; 1: handed back
// -- outer iteration *it_io is the caller's, from pose *T_io; 5: a wait for a peer ran out (every rank reports it), the
// connection's inboxes carry a raised abort word and the caller serves everything else without them.
int pipe_run(PipeRank *rk, int nranks, int world, size_t n_total, Pose *T_io, size_t *it_io, size_t max_iter,
             uint32_t *inner_iters, int *why) {
  *why = 1;
  size_t it = *it_io;
  Pose T = *T_io;
  if (it >= max_iter) {
    *why = 0;
    return ICP_OK;
  }
  if (nranks < 1 || world < 1 || world > kShardMaxWorld || n_total < ((size_t)1 << 12)) return ICP_OK;
  int B, threads;
  reduce_geometry(n_total, &B, &threads);
  if ((size_t)B * (size_t)threads * 8 < n_total) return ICP_OK;  // (beyond 2^24 points a thread folds more than eight)
  for (int j = 0; j < nranks; ++j) {
    icp_handle *h = rk[j].h;
    Workspace &w = h->ws;
    if (!w.d_loop_inbox || w.loop_rank != rk[j].rank || w.loop_world != world || rk[j].nbl < 1 || rk[j].nbl > kReduceMaxBlocks ||
        rk[j].n_local == 0 || resolved_nn_mode(h) != ICP_NN_GRID)
      return ICP_OK;
    const QuerySort &Q = h->qsort;
    if (!(Q.valid && Q.src == rk[j].d_src && Q.n == rk[j].n_local)) return ICP_OK;  // (the snapshot the searches below work in)
  }
  {
    Workspace &w0 = rk[0].h->ws;
    if (w0.pipe_off > 0) {  // (replicated: every rank counts the same misses)
      for (int j = 0; j < nranks; ++j) --rk[j].h->ws.pipe_off;
      return ICP_OK;
    }
  }
  // the kinds of an iteration's two evaluations (common.hpp: Workspace::win_kind): a call's first iteration is predicted
  // from the previous call's first iteration (3, 4), every other from the iteration before it (0, 1)
  auto kind1 = [](size_t i) { return i == 0 ? 3 : 0; };
  auto kind2 = [](size_t i) { return i == 0 ? 4 : 1; };
  PipeWin W1, W2;
  if (!pipe_window(rk[0].h, n_total, kind1(it), &W1) || !pipe_window(rk[0].h, n_total, kind2(it), &W2)) return ICP_OK;
  for (int j = 1; j < nranks; ++j) {  // ranks of one process: their histories are the same history
    PipeWin a, b;
    if (!pipe_window(rk[j].h, n_total, kind1(it), &a) || !pipe_window(rk[j].h, n_total, kind2(it), &b) ||
        memcmp(&a.P, &W1.P, sizeof(WinParams)) != 0 || memcmp(&b.P, &W2.P, sizeof(WinParams)) != 0)
      return ICP_OK;
  }
  for (int j = 0; j < nranks; ++j) {
    icp_handle *h = rk[j].h;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(ensure_workspace(h, rk[j].n_local, false));
    Workspace &w = h->ws;
    for (int c = 0; c < 2; ++c) {  // both evaluation contexts in their rest state
      if (w.gn_dirty) {
        HIP_TRY(launch_sel_init(h, rk[j].n_local));
        w.gn_dirty = false;
      }
      w.swap_ctx();
    }
  }

  auto pairs_a = [&](int j, int k) { Workspace &w = rk[j].h->ws; return k == 0 ? w.d_a : (k == 1 ? w.d_a2 : w.d_a3); };
  auto pairs_b = [&](int j, int k) { Workspace &w = rk[j].h->ws; return k == 0 ? w.d_b : (k == 1 ? w.d_b2 : w.d_b3); };
  // the search of outer iteration `for_it` (it reports the correspondences if it is the call's last)
  auto search_all = [&](const Pose &pose, int k, size_t for_it) -> int {
    for (int j = 0; j < nranks; ++j) {
      uint32_t *idx = (for_it + 1 == max_iter) ? rk[j].d_idx : nullptr;
      const int rc = icp_correspond_device(rk[j].h, rk[j].d_src, rk[j].n_local, &pose, pairs_a(j, k), pairs_b(j, k), idx);
      if (rc != ICP_OK) return rc;
    }
    return ICP_OK;
  };
  // ... with its pose read from device memory, where the finishing workgroup of the evaluation in front of it leaves it
  auto ahead_all = [&](int k, size_t for_it, bool *issued) -> int {
    *issued = true;
    for (int j = 0; j < nranks; ++j) {
      icp_handle *h = rk[j].h;
      HIP_TRY(hipSetDevice(h->device));
      uint32_t *idx = (for_it + 1 == max_iter) ? rk[j].d_idx : nullptr;
      bool launched = false;
      HIP_TRY(launch_nn_grid_ahead(h, rk[j].d_src, rk[j].n_local, h->ws.d_ahead, pairs_a(j, k), pairs_b(j, k), idx, &launched));
      if (!launched) *issued = false;  // (cannot differ between ranks: the same kind of snapshot everywhere)
    }
    return ICP_OK;
  };
  // one or two evaluations on every local rank: ranks of one device (consecutive in rk) in one finishing launch
  auto launch_evals = [&](const ShardPickEval *ev, int nevals, const int *buf_of_eval) -> int {
    const unsigned gen0 = rk[0].h->ws.pipe_gen + 1u;
    for (int j0 = 0; j0 < nranks;) {
      int j1 = j0 + 1;
      while (j1 < nranks && j1 - j0 < 8 && rk[j1].h->device == rk[j0].h->device && rk[j1].h->stream == rk[j0].h->stream) ++j1;
      ShardPickRank sr[8];
      for (int j = j0; j < j1; ++j) {
        ShardPickRank &s = sr[j - j0];
        s.h = rk[j].h;
        s.rank = rk[j].rank;
        s.b0 = rk[j].b0;
        s.nbl = rk[j].nbl;
        s.n_local = rk[j].n_local;
        for (int e = 0; e < nevals; ++e) {
          s.a[e] = pairs_a(j, buf_of_eval[e]);
          s.b[e] = pairs_b(j, buf_of_eval[e]);
        }
      }
      HIP_TRY(hipSetDevice(rk[j0].h->device));
      HIP_TRY(launch_shard_evals(sr, j1 - j0, world, B, n_total, gen0, ev, nevals));
      j0 = j1;
    }
    for (int j = 0; j < nranks; ++j) rk[j].h->ws.pipe_gen += (unsigned)nevals;
    return ICP_OK;
  };
  // the result of the evaluation in context `alt`: rank 0's copy (every rank releases the same bits; checked)
  auto wait_eval = [&](bool alt, GnResult *out) -> int {
    for (int j = 0; j < nranks; ++j) {
      icp_handle *h = rk[j].h;
      GnCtx &c = alt ? h->ws.alt : static_cast<GnCtx &>(h->ws);
      HIP_TRY(hipSetDevice(h->device));
      HIP_TRY(wait_seq(h, &c.h_res->seq, c.seq, h->stream));
      if (j == 0) {
        *out = *c.h_res;
      } else if (memcmp(out->acc, c.h_res->acc, sizeof(out->acc)) != 0 || out->overflow != c.h_res->overflow ||
                 out->nan_flag != c.h_res->nan_flag || memcmp(out->sigma, c.h_res->sigma, sizeof(out->sigma)) != 0) {
        return ICP_HIP_ERROR;  // (cannot happen: the ranks fold the same numbers in the same order)
      }
    }
    return ICP_OK;
  };
  auto quiesce = [&]() {
    for (int j = 0; j < nranks; ++j) {
      (void)hipSetDevice(rk[j].h->device);
      (void)hipStreamSynchronize(rk[j].h->stream);
    }
  };
  // a miss, as wgn_step books it: wider windows for that kind (2), or a pause for the files (3)
  auto book_miss = [&](int kind, int overflow) {
    for (int j = 0; j < nranks; ++j) {
      Workspace &w = rk[j].h->ws;
      if (overflow == 3) {
        ++w.bkt_misses;
        w.pipe_off = 16;
      } else {
        ++w.win_missed;
        w.win_kind[kind].wide = true;
      }
    }
  };
  auto leave = [&](int reason) -> int {
    quiesce();
    for (int j = 0; j < nranks; ++j) {
      Workspace &w = rk[j].h->ws;
      if (reason == 5) ++w.pipe_gave_up;
      else if (reason == 1) ++w.pipe_handbacks;
    }
    *why = reason;
    *it_io = it;
    *T_io = T;
    return ICP_OK;
  };

  Range range("icp: pipelined sharded iterations (search -> paired first launches -> finishing workgroups across ranks)");
  int cur = 0;
  ICP_TRY_RC(search_all(T, cur, it));
  {
    ShardPickEval e1 = {};
    e1.alt_ctx = false;
    e1.T = transform_identity();
    e1.P = W1.P;
    e1.ahead_on = it + 1 < max_iter;
    e1.outer = T;
    const int bufs[1] = {cur};
    for (int j = 0; j < nranks; ++j) ++rk[j].h->ws.win_tried;
    ICP_TRY_RC(launch_evals(&e1, 1, bufs));
  }
  bool ahead_issued = false;
  if (it + 1 < max_iter) ICP_TRY_RC(ahead_all((cur + 1) % 3, it + 1, &ahead_issued));
  GnResult r1;
  ICP_TRY_RC(wait_eval(false, &r1));
  for (;;) {
    // r1: the first evaluation of outer iteration `it` (inner pose = identity), pairs in buffer `cur`
    if (r1.overflow == 5) return leave(5);
    if (r1.nan_flag) {
      for (int j = 0; j < nranks; ++j) rk[j].h->ws.gn_dirty = true;
      return leave(1);  // (the caller's own evaluation reports the NaN)
    }
    if (r1.overflow) {
      book_miss(kind1(it), r1.overflow);
      return leave(1);
    }
    for (int j = 0; j < nranks; ++j) pipe_record(rk[j].h->ws, kind1(it), W1, r1);
    double delta1[3];
    if (!solve_update(r1.acc, r1.acc + 9, delta1)) return leave(1);           // src/lib.rs:67-69
    if (norm2(delta1) < ICP_DELTA_NORM_THRESHOLD) return leave(1);            // :71-73 (no update: not the steady state)
    const double err1 = r1.acc[12];                                            // (never > f64::MAX: :75-78)
    const Pose T1 = transform_mul(transform_new(delta1), transform_identity());  // :81
    const Pose spec = transform_mul(T1, T);                                       // :127, 170 -- if the deciding evaluation ends the loop
    const bool last = it + 1 == max_iter;
    if (!pipe_window(rk[0].h, n_total, kind2(it), &W2)) return leave(1);
    const int nxt = (cur + 1) % 3, nxt2 = (cur + 2) % 3;
    for (int j = 0; j < nranks; ++j) ++rk[j].h->ws.win_tried;
    if (last) {
      ShardPickEval e2 = {};
      e2.alt_ctx = true;
      e2.T = T1;
      e2.P = W2.P;
      const int bufs[1] = {cur};
      ICP_TRY_RC(launch_evals(&e2, 1, bufs));
    } else {
      if (!pipe_window(rk[0].h, n_total, kind1(it + 1), &W1)) return leave(1);
      // the search for `spec`: in flight already if the device derived the same pose
      const bool have_search = ahead_issued && r1.next_valid != 0 && memcmp(&r1.next_pose, &spec, sizeof(Pose)) == 0;
      for (int j = 0; j < nranks && ahead_issued; ++j) ++(have_search ? rk[j].h->ws.ahead_hits : rk[j].h->ws.ahead_misses);
      if (!have_search) ICP_TRY_RC(search_all(spec, nxt, it + 1));
      ShardPickEval ev[2] = {};
      ev[0].alt_ctx = false;  // the NEXT iteration's first evaluation ...
      ev[0].T = transform_identity();
      ev[0].P = W1.P;
      ev[0].ahead_on = it + 2 < max_iter;
      ev[0].outer = spec;
      ev[1].alt_ctx = true;   // ... beside this iteration's deciding one
      ev[1].T = T1;
      ev[1].P = W2.P;
      const int bufs[2] = {nxt, cur};
      for (int j = 0; j < nranks; ++j) {
        ++rk[j].h->ws.win_tried;
        ++rk[j].h->ws.pre_evals;
      }
      ICP_TRY_RC(launch_evals(ev, 2, bufs));
      ahead_issued = false;
      if (it + 2 < max_iter) ICP_TRY_RC(ahead_all(nxt2, it + 2, &ahead_issued));
    }
    GnResult r2;
    ICP_TRY_RC(wait_eval(true, &r2));
    if (r2.overflow == 5) return leave(5);
    if (r2.nan_flag) {
      for (int j = 0; j < nranks; ++j) rk[j].h->ws.gn_dirty = true;
      return leave(1);
    }
    if (r2.overflow) {
      book_miss(kind2(it), r2.overflow);
      return leave(1);
    }
    for (int j = 0; j < nranks; ++j) pipe_record(rk[j].h->ws, kind2(it), W2, r2);
    double delta2[3];
    const bool stop = !solve_update(r2.acc, r2.acc + 9, delta2) || norm2(delta2) < ICP_DELTA_NORM_THRESHOLD || r2.acc[12] > err1;
    if (!stop) {  // the inner loop goes on: the bet is off, the caller's loop serves this iteration from its start
      for (int j = 0; j < nranks; ++j) ++rk[j].h->ws.spec_misses;
      return leave(1);
    }
    if (inner_iters) inner_iters[it] = 1u;
    T = spec;
    ++it;
    for (int j = 0; j < nranks; ++j) {
      Workspace &w = rk[j].h->ws;
      ++w.pipe_iters;
      ++w.spec_hits;
      w.last_inner = 1u;
    }
    if (last) return leave(0);
    cur = nxt;
    ICP_TRY_RC(wait_eval(false, &r1));  // (launched beside the deciding evaluation)
  }
}

}  // namespace icp

// One process per GPU: this rank's iterations of the pipelined sharded registration (include/icp_mi355x.h section 5c).
// d_src_local: the rank's points (icp_shard_take_device out of the fold order), its search snapshot prepared
// (icp_prepare_source_device) and searched at least once (a slow iteration precedes: it seeds the window predictions).
// Every rank of the connection calls this at the same point of the registration with the same arguments and takes the
// same decisions from the same bits; *why as pipe_run's.
extern "C" int icp_shard_pipe_run_device(icp_handle *h, const double *d_src_local, size_t n_local, size_t n_total, int rank, int world,
                                         icp_pose *T_io, size_t *it_io, size_t max_iter, uint32_t *inner_iters,
                                         uint32_t *d_idx_local, int *why) {
  if (!h || !T_io || !it_io || !why || world < 1 || world > kShardMaxWorld || rank < 0 || rank >= world || n_total >= 0xffffffffull ||
      (n_local > 0 && !d_src_local))
    return ICP_BAD_ARGUMENT;
  int b0, b1, B;
  size_t nl;
  shard_geometry(n_total, rank, world, &b0, &b1, &B, &nl);
  if (nl != n_local) return ICP_BAD_ARGUMENT;
  PipeRank rk = {h, d_src_local, n_local, rank, b0, b1 - b0, d_idx_local};
  return pipe_run(&rk, 1, world, n_total, T_io, it_io, max_iter, inner_iters, why);
}

// out[0] outer iterations the pipeline served on this handle, [1] times it handed back, [2] times it gave up waiting for
// a peer, [3] run-ahead searches whose pose the host confirmed
extern "C" int icp_pipe_counters(icp_handle *h, uint64_t out[4]) {
  if (!h || !out) return ICP_BAD_ARGUMENT;
  out[0] = h->ws.pipe_iters;
  out[1] = h->ws.pipe_handbacks;
  out[2] = h->ws.pipe_gave_up;
  out[3] = h->ws.ahead_hits;
  return ICP_OK;
}
