/*
 * icp_mi355x.h -- C ABI of the MI355X (gfx950) ICP registration core.
 *
 * Drop-in boundary for the hot path of tier4/icp_rust: the reference has no FFI of its
 * own; its boundary is the crate's public Rust API (src/lib.rs:12-26).  Each entry
 * point below names the Rust item (file:line under /root/reference) it stands in for.
 * INTEGRATION.md shows the `extern "C"` block + safe wrappers a maintainer adds to the
 * crate so that `Icp2d`, `Icp3d`, `Transform`, `estimate_transform`, ... keep their
 * signatures.  Plain pointers and sizes only; no C++ or torch types cross this line.
 *
 * Conventions
 *  - points are dense AoS doubles exactly as `&[Vector2]` / `&[Vector3]` lie in memory
 *    (nalgebra ArrayStorage<f64, D, 1>: stride 16 B / 24 B), src/types.rs:4-12;
 *  - `icp_pose` is `Transform { rot: Rotation2, t: Vector2 }` (src/transform.rs:6-10),
 *    rotation column-major as nalgebra stores it;
 *  - every function returns an icp_status; the reference's `None` is ICP_NONE, its two
 *    panics (empty dst: src/lib.rs:122,165; NaN residual: src/stats.rs:12) are
 *    ICP_EMPTY_DST / ICP_NAN_INPUT;
 *  - one in-flight call per handle (it owns scratch + a HIP stream); handles are
 *    independent.  Entry points that take DEVICE pointers run on the handle's private,
 *    non-blocking streams unless icp_set_stream was called: the buffers they read must be
 *    complete when the call is made (synchronise the stream that produced them, or put the
 *    handle on that stream with icp_set_stream -- then everything is ordered there).  When a
 *    call returns, whatever its status, none of its work is still in flight on the caller's
 *    buffers.  All compute runs on the GPU: without a usable HIP device every
 *    compute entry point returns ICP_NO_DEVICE -- there is no CPU fallback.
 */
#ifndef ICP_MI355X_H
#define ICP_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: additions only (sharded stage calls, icp_multi, single-launch switch, point-to-plane extension, icp_f64_sin/cos);
 * everything of version 1 is unchanged */
/* 3: additions only (icp_last_fold_order; icp_sort_source_device).  Behavioural note: icp_estimate[_device] on the grid
 * engine now folds its sums over the source cloud in FOLD ORDER (a deterministic sort by target-grid cell, section 9a)
 * instead of the caller's order -- same correspondences, pose equal within the rounding of a re-ordered sum (~1e-12) */
/* 4: the sharded evaluation has TWO exchanges: icp_shard_eval_accumulate_device is gone, icp_shard_eval_compact_device
 * hands out one block per rank (candidates + block sums, icp_shard_exchange_bytes) and icp_shard_eval_finish_device
 * takes the gathered blocks.  Additions: icp_shard_exchange_bytes, icp_nn_cert_counters, icp_multi_append_targets,
 * icp_grid_append_counters.  Behavioural note: the sums of the weighted normal equations are kept per dimension and
 * scaled by 1 / sigma after the fold (g_x S_x + g_y S_y) instead of per term -- pose equal within rounding (~1e-13) */
/* 5: additions only (section 5b: the one-launch sharded inner loop -- icp_loop_inbox*, icp_loop_ipc_*,
 * icp_shard_loop_*; icp_multi_compute/update_target_normals, icp_multi_estimate_point_to_plane).  The counters and
 * icp_profile_* moved to icp_mi355x_debug.h (same symbols, same library).  Behavioural note: clouds of at most 2^20
 * points run their inner loop inside one launch; results are the same bits as the stepped loop's */
/* 6: icp_loop_inbox takes the KIND of memory (ICP_INBOX_*: 0 and 1 are what `fine_grained` 0 / 1 meant); additions:
 * ICP_INBOX_HOST and icp_loop_inbox_shm_name / _shm_unlink / icp_loop_shm_open / _shm_close (inboxes in pinned host
 * memory shared between processes), icp_loop_transport_probe, icp_reset_window_predictions.  Behavioural notes: a
 * sharded one-launch inner loop that gives up raises the abort word in every rank's inbox, so all ranks report
 * ICP_HIP_ERROR for that launch; the evaluations of icp_estimate[_device] file their candidates in the first pass over
 * the pairs and finish in one workgroup (same bits: the order statistics are exact either way) */
/* 7: additions only (section 5c: icp_shard_pipe_run_device, the pipelined sharded registration; icp_pipe_counters,
 * icp_multi_pipe_iterations in icp_mi355x_debug.h).  Behavioural note: icp_multi_estimate and the sharded drivers serve
 * the steady state of a registration through it (same bits) */
#define ICP_ABI_VERSION 7

typedef enum icp_status {
  ICP_OK = 0,
  ICP_NONE = 1,        /* Option::None (src/lib.rs:67-69, 186-189, 257-260)           */
  ICP_EMPTY_DST = 2,   /* reference panics: index.unwrap(), src/lib.rs:122,165        */
  ICP_NAN_INPUT = 3,   /* reference panics: partial_cmp().unwrap(), src/stats.rs:12   */
  ICP_BAD_ARGUMENT = 4,
  ICP_NO_DEVICE = 5,   /* HIP runtime/device unavailable -- no CPU fallback exists    */
  ICP_HIP_ERROR = 6,
  ICP_OUT_OF_MEMORY = 7,
  ICP_RETRY_REPLICATED = 8, /* sharded evaluation (section 5): evaluate this one on the gathered pairs */
  ICP_RETRY_SHARDED = 9     /* sharded evaluation: the window missed; hist again with refined = 1      */
} icp_status;

/* Transform (src/transform.rs:6-10) */
typedef struct icp_pose {
  double r00, r10, r01, r11; /* Rotation2, column-major */
  double tx, ty;             /* Vector2                 */
} icp_pose;

/* constants of the path (read-only; src/lib.rs:32,60,61, src/stats.rs:42) */
#define ICP_HUBER_K 1.345
#define ICP_DELTA_NORM_THRESHOLD 1e-6
#define ICP_INNER_MAX_ITER 200
#define ICP_PPF34 1.482602218505602

/* nearest-neighbour engines: identical results (exact NN, d^2 = ((dx^2+dy^2)+dz^2)
 * without FMA, ties -> lowest index), different cost. */
typedef enum icp_nn_mode {
  ICP_NN_AUTO = 0,
  ICP_NN_BRUTE = 1, /* LDS-tiled brute force, O(N*M) f64                              */
  ICP_NN_GRID = 2   /* exact uniform-grid search built on device at icp_create        */
} icp_nn_mode;

const char *icp_status_string(int status);
int icp_abi_version(void);
/* number of usable HIP devices (0 on a CPU-only host; never initialises a context) */
int icp_device_count(void);
/* the PCI bus id of a device ("0000:c1:00.0"): what tells two devices of one node apart whatever ordinal a process sees
 * them under (the sharded drivers pick the transport of their inboxes by it) */
int icp_device_pci_bus_id(int device, char out[64]);

/* ================================================================================
 * 1. Host-side pose algebra: Transform / se2 / so2 (tiny, runs on the host exactly as
 *    in the reference; exported so the host mirror and the tests share one definition)
 * ============================================================================== */
void icp_transform_new(const double param[3], icp_pose *out);      /* Transform::new, transform.rs:13-16 -> se2::calc_rt se2.rs:21-41 */
void icp_transform_from_rt(const double rot_colmajor[4], const double t[2], icp_pose *out); /* transform.rs:18-20 */
void icp_transform_identity(icp_pose *out);                        /* transform.rs:34-39 */
void icp_transform_apply(const icp_pose *T, const double p[2], double out[2]);   /* Transform::transform, transform.rs:22-24 */
void icp_transform_inverse(const icp_pose *T, icp_pose *out);      /* transform.rs:26-32 */
void icp_transform_mul(const icp_pose *lhs, const icp_pose *rhs, icp_pose *out); /* impl Mul, transform.rs:42-51 */
void icp_se2_exp(const double param[3], double m3_rowmajor[9]);    /* se2::exp, se2.rs:43-52 */
void icp_se2_log(const double m3_rowmajor[9], double param[3]);    /* se2::log, se2.rs:54-77 */
void icp_se2_get_rt(const double m3_rowmajor[9], double rot_rowmajor[4], double t[2]); /* se2::get_rt, se2.rs:11-19 */
void icp_so2_exp(double theta, double m2_colmajor[4]);             /* so2::exp / new_rotation2, so2.rs:8-31 */
double icp_so2_log(const double m2_colmajor[4]);                   /* so2::log, so2.rs:19-21 */
double icp_norm(const double *m_colmajor, size_t nrows, size_t ncols); /* icp::norm, norm.rs:19-21 */
int icp_inverse3x3(const double m_rowmajor[9], double out_rowmajor[9]); /* linalg::inverse3x3, linalg.rs:3-29 (ICP_NONE iff det == 0) */
/* f64::sin / f64::cos as the reference's no_std build evaluates them (num-traits `libm` feature,
 * Cargo.toml:17-20 -> the libm crate, a port of musl's kernels; restated in icp_trig.h).  Host and
 * device share the definition: the inner loop's pose updates (Transform::new, src/lib.rs:81) run
 * on the GPU and must give the bits the host API gives. */
double icp_f64_sin(double x);
double icp_f64_cos(double x);
/* Transform::new on the DEVICE for n parameter vectors (host buffers in and out; test / observability
 * entry: proves that device and host evaluate se2::calc_rt to the same bits) */
int icp_transform_new_device(const double *params_xyz, size_t n, icp_pose *out, int device);

/* ================================================================================
 * 2. The registration handle: Icp2d / Icp3d
 * ============================================================================== */
typedef struct icp_handle icp_handle;

/* Icp2d::new (src/lib.rs:97-102) / Icp3d::new (src/lib.rs:139-144).  dim = 2 | 3.
 * Copies `dst` (m points, AoS) to the device and builds the search structure there, so
 * `dst` may be freed afterwards (stricter than the reference's borrow `'a`, never
 * weaker).  device < 0 selects the current HIP device. */
int icp_create(icp_handle **out, int dim, const double *dst, size_t m, int device);
/* same, but `d_dst` already lives in device memory (m x dim doubles, AoS); it is
 * borrowed for the lifetime of the handle, like the reference's `&'a [Vector]`. */
int icp_create_device(icp_handle **out, int dim, const double *d_dst, size_t m, int device);
void icp_destroy(icp_handle *h);

/* Clouds of the reference's own size (scans/2d: ~650 points; here: up to 1024 source points against up
 * to 2048 targets) are registered by icp_estimate[_device] in ONE launch of one workgroup -- search,
 * inner loop, 3x3 solve, break tests and pose updates all on the device, same bits as the general
 * path.  icp_set_single_launch(h, 0) switches that off for a handle (tests, A/B); the counters:
 * out[0] calls served, out[1] Gauss-Newton evaluations in them, out[2] of which needed the sorting path. */
int icp_set_single_launch(icp_handle *h, int enable);
int icp_set_nn_mode(icp_handle *h, int mode);   /* icp_nn_mode; default ICP_NN_AUTO */
int icp_get_nn_mode(const icp_handle *h);       /* the engine AUTO resolved to      */
/* run the handle's kernels on a caller-chosen HIP stream (hipStream_t) instead of the handle's
 * own non-blocking stream.  NULL is the HIP default (null) stream -- what torch.cuda's default
 * stream is -- not "reset".  Used by hosts that interleave their own device work (e.g. RCCL
 * collectives, torch ops) with the stage-level calls of section 4: everything is then ordered
 * on that one stream.  icp_use_own_stream goes back to the private stream. */
int icp_set_stream(icp_handle *h, void *hip_stream);
int icp_use_own_stream(icp_handle *h);

/* Icp2d::estimate (src/lib.rs:105-130) / Icp3d::estimate (src/lib.rs:148-173):
 * the result of exactly `max_iter` outer iterations of transform -> exact NN -> estimate_transform
 * -> compose.  (An iteration that applies no update and returns the pose it was given, bit for bit, is a fixed point:
 * the iterations after it would repeat it and are not run, except the last, which reports the correspondences; their
 * inner counts are 0, as they would be.  icp_fixed_point_skips, icp_mi355x_debug.h, counts them.)
 * src: n points AoS (host).  Optional outputs (NULL to skip):
 *   last_idx[n]          correspondence indices of the last outer iteration,
 *   inner_iters[max_iter] inner Gauss-Newton updates applied per outer iteration. */
int icp_estimate(icp_handle *h, const double *src, size_t n, const icp_pose *init,
                 size_t max_iter, icp_pose *out, uint32_t *last_idx, uint32_t *inner_iters);
/* same with `d_src` (and optional d_last_idx) resident in device memory */
int icp_estimate_device(icp_handle *h, const double *d_src, size_t n, const icp_pose *init,
                        size_t max_iter, icp_pose *out, uint32_t *d_last_idx,
                        uint32_t *inner_iters);

/* ================================================================================
 * 3. The robust pose estimator as free functions (host buffers; a, b are n x 2 AoS)
 * ============================================================================== */
/* icp::estimate_transform, src/lib.rs:59-84.  inner_iters (nullable) = updates applied */
int icp_estimate_transform(const double *a_xy, const double *b_xy, size_t n, icp_pose *out,
                           uint32_t *inner_iters);
/* icp::weighted_gauss_newton_update, src/lib.rs:218-261 (ICP_NONE where it returns None) */
int icp_weighted_gauss_newton_update(const icp_pose *T, const double *a_xy, const double *b_xy,
                                     size_t n, double delta[3]);
/* icp::gauss_newton_update, src/lib.rs:191-216 */
int icp_gauss_newton_update(const icp_pose *T, const double *a_xy, const double *b_xy, size_t n,
                            double delta[3]);
/* icp::error, src/lib.rs:38-43 and icp::huber_error, src/lib.rs:45-50 */
int icp_error(const icp_pose *T, const double *a_xy, const double *b_xy, size_t n, double *out);
int icp_huber_error(const icp_pose *T, const double *a_xy, const double *b_xy, size_t n, double *out);
/* stats::calc_stddevs over the residuals T*a - b (src/stats.rs:49-60, called at
 * src/lib.rs:236): sigma[2] = 1.4826 * MAD per dimension.  Exposed for parity tests. */
int icp_residual_stddevs(const icp_pose *T, const double *a_xy, const double *b_xy, size_t n,
                         double sigma[2]);

/* ================================================================================
 * 4. Stage-level calls on device memory (what a multi-GPU host composes: shard the
 *    source cloud over ranks, all-gather the matched pairs, replicate the tiny solve)
 * ============================================================================== */
/* stage (i) of one outer iteration, src/lib.rs:113-124 / 156-167 (+ get_xy :86-89):
 * s' = T (.) src; idx = NN(s'); a = xy(s'); b = xy(dst[idx]).  d_src: n x dim AoS;
 * d_a, d_b: n x 2 AoS out; d_idx: n out (nullable).  Asynchronous on the handle's
 * stream. */
int icp_correspond_device(icp_handle *h, const double *d_src, size_t n, const icp_pose *T,
                          double *d_a_xy, double *d_b_xy, uint32_t *d_idx);
/* a = xy(T (.) src), b = xy(dst[idx]) for n points from given correspondence indices -- the
 * second half of icp_correspond_device on its own.  A multi-GPU host that replicates the source
 * cloud all-gathers only the 4-byte indices of each rank's shard (instead of 32 bytes of pairs
 * per point) and rebuilds the pairs locally with this call; same arithmetic, same bits. */
int icp_materialize_pairs_device(icp_handle *h, const double *d_src, size_t n, const icp_pose *T,
                                 const uint32_t *d_idx, double *d_a_xy, double *d_b_xy);
/* Optional, before a run of icp_correspond_device calls on the same source buffer: takes a
 * cell-sorted snapshot of (d_src, n) under pose T so that neighbouring GPU lanes search
 * neighbouring cells (results are unchanged, bit for bit; only memory locality changes).
 * The snapshot is used by later icp_correspond_device calls with the same (d_src, n) and
 * must be retaken if the buffer's contents change.  icp_estimate* do this themselves. */
int icp_prepare_source_device(icp_handle *h, const double *d_src, size_t n, const icp_pose *T);
/* stage (ii)+(iii): the whole inner loop of src/lib.rs:59-84 on device-resident pairs;
 * synchronises the stream (the 3x3 solve and the break tests run on the host). */
int icp_estimate_transform_device(icp_handle *h, const double *d_a_xy, const double *d_b_xy,
                                  size_t n, icp_pose *out, uint32_t *inner_iters);
/* brute-force / grid NN alone (parity tests): d_q n x dim AoS -> d_idx[n] */
int icp_nn_search_device(icp_handle *h, const double *d_q, size_t n, uint32_t *d_idx);
int icp_synchronize(icp_handle *h);

/* One evaluation of weighted_gauss_newton_update (src/lib.rs:218-261) plus the Huber error of the same
 * pose (:75) on device pairs -- the body of one inner iteration of icp_estimate_transform_device, for
 * hosts that drive the inner loop themselves.  kind: 0 = first evaluation on new correspondences,
 * 1 = the evaluation after the first update, 2 = later ones (only selects which earlier evaluation
 * predicts this one's statistics; results never depend on it).  ICP_NONE where the reference returns
 * None. */
int icp_weighted_gn_step_device(icp_handle *h, const double *d_a_xy, const double *d_b_xy, size_t n,
                                const icp_pose *T, int kind, double delta[3], double *huber_err);

/* ================================================================================
 * 5. Sharded evaluation: the inner loop's sums across the GPUs of a node, bit-identical to one GPU
 * ==============================================================================
 * SURVEY.md 8(e).  The N-term sums are folded in a fixed tree of `blocks x 512` threads
 * (icp_reduce_geometry).  Rank r of `world` owns the blocks [blocks r / world, blocks (r+1) / world):
 * the points those blocks fold (icp_shard_geometry: n_local of them; icp_shard_take_device compacts
 * them out of a full array in fold order, icp_shard_put_device is the inverse).  It searches their
 * nearest neighbours (icp_correspond_device on its compact source cloud) and evaluates them with
 *   icp_shard_eval_hist_device        residuals + window histograms of its points   -> *d_hist
 *                                     (+ the running sums of its blocks, kept in the handle: since ABI 4 they
 *                                     carry no 1 / sigma and so need not wait for the statistics)
 *      [host: SUM the icp_shard_histogram_words() u32 at *d_hist over all ranks, in place -- EVERY rank, whatever
 *       its hist call answered (short of ICP_BAD_ARGUMENT): the last four words are status counters, one-hot per
 *       rank {OK, RETRY_REPLICATED, NONE, other}; after the sum every rank reads the same four counts
 *       (icp_shard_eval_status) and takes the same branch, also when a rank-local condition made one rank answer
 *       differently from its peers -- the ranks must never enter different collectives]
 *   icp_shard_eval_compact_device     its candidates around the median / the MAD
 *                                     and its block sums                             -> d_exchange_out
 *      [host: ALL-GATHER icp_shard_exchange_bytes(world) bytes per rank, rank order]
 *   icp_shard_eval_finish_device      exact statistics from all candidates, second stage over all block
 *                                     sums, 3x3 solve                                -> delta, Huber error
 * Two exchanges per evaluation (ABI 3 had a third: icp_shard_eval_accumulate_device between compact and finish).
 * What crosses ranks is integer counts, order statistics candidates and per-block sums placed in block
 * order, so every rank ends with the bits one GPU computes.  The exchange is the host's: RCCL
 * (icp_rust_amd/dist.py drives these calls through torch.distributed), or the peer copies of
 * icp_multi_* below.  ICP_RETRY_REPLICATED (from hist: no prediction of this evaluation's statistics
 * yet / fewer blocks than ranks; from finish: the predicted window missed) means: gather the pairs of
 * all ranks in global order and call icp_weighted_gn_step_device on them instead -- same result, and
 * it provides the prediction.  ICP_RETRY_SHARDED from finish: the window missed, but its (exact,
 * global) counts place a narrower one that will not -- run the three stages again with refined = 1.
 * All calls are asynchronous on the handle's stream except finish. */
int icp_shard_geometry(size_t n_total, int rank, int world, int *block_first, int *block_end, int *blocks,
                       size_t *n_local);
size_t icp_shard_histogram_words(void);
size_t icp_shard_candidates_bytes(void);     /* layout of the exchanged block: the candidates ... */
size_t icp_shard_partials_bytes(int world);  /* ... then the block sums */
size_t icp_shard_exchange_bytes(int world);  /* = the two together */
int icp_shard_take_device(icp_handle *h, const void *d_full, void *d_local, size_t n_total, int rank, int world,
                          size_t elem_bytes);
/* icp_sort_source_device + icp_shard_take_device in one: the fold order of the WHOLE cloud under pose T and this rank's
 * points gathered through it, without a sorted copy of the whole cloud.  d_perm: nullable (n_total words). */
int icp_shard_sort_take_device(icp_handle *h, const double *d_src_full, size_t n_total, const icp_pose *T, int rank, int world,
                               double *d_local, uint32_t *d_perm);
/* icp_prepare_source_device for a rank's points as icp_shard_take_device compacts them out of the fold order: runs of
 * the cell-sorted cloud, whose order the search snapshot keeps (no second sort) */
int icp_shard_prepare_source_device(icp_handle *h, const double *d_src_local, size_t n_local, const icp_pose *T);
int icp_shard_put_device(icp_handle *h, const void *d_local, void *d_full, size_t n_total, int rank, int world,
                         size_t elem_bytes);
int icp_shard_eval_hist_device(icp_handle *h, const double *d_a_xy_local, const double *d_b_xy_local, size_t n_total,
                               int rank, int world, const icp_pose *T, int kind, int refined, uint32_t **d_hist);
int icp_shard_eval_compact_device(icp_handle *h, void *d_exchange_out);
int icp_shard_eval_finish_device(icp_handle *h, const void *d_exchange_all, double delta[3], double *huber_err);
/* the four status counts of the evaluation in flight.  from_device = 0: as finish left them in host memory (valid once
 * finish has returned, no wait); 1: read from the summed buffer behind the stream (a rank whose own hist call was not
 * ICP_OK and which ran no finish).  icp_shard_eval_abort_device: after an evaluation the ranks did not ALL answer with
 * ICP_OK, back to the rest state the next evaluation expects (a rank without a histogram of its own holds its peers'
 * counts after the sum). */
int icp_shard_eval_status(icp_handle *h, uint32_t out[4], int from_device);
int icp_shard_eval_abort_device(icp_handle *h);

/* ---- 5b. The inner loop of a sharded registration in ONE launch per rank (estimate_transform, src/lib.rs:59-84) ----
 * Round 4.  The stage calls above cost ten enqueues and a host wait per evaluation and rank.  Here a rank launches its
 * tree blocks' workgroups ONCE per outer iteration; they keep the rank's pairs on chip, run evaluation after evaluation
 * and exchange the window histograms, the candidates and the block sums with the other ranks THROUGH MEMORY while they
 * run: every rank owns an INBOX (icp_loop_inbox, icp_loop_inbox_bytes() of device memory) that every peer has mapped --
 * plain pointers between ranks of one process (peer access between devices), hipIpc handles between processes
 * (icp_loop_inbox_ipc_handle / icp_loop_ipc_open) -- a producer writes its bytes into every inbox and then a flag word,
 * a consumer polls and reads only its own.  3 x 3 solve, break tests and Transform::new run on the device (every rank
 * computes the same bits); the host sees one result per launch.  Results: the bits of one handle, as with the stage calls.
 *   icp_shard_loop_connect: every rank's inbox as mapped in this process (inboxes[rank] = this handle's own); empties
 *     the own inbox, so the ranks must meet (any barrier of the driver) between connecting and the first launch.
 *   icp_shard_loop_launch_device: from evaluation it0 with the loop's state (inner pose, previous Huber error, updates
 *     applied: src/lib.rs:62-82); launch_no = 1, 2, ... the same on every rank for the same launch; eval_base = the
 *     evaluations earlier launches of this connection served.  ICP_RETRY_SHARDED: nothing launched (no window
 *     prediction for it0 yet, more than 2^23 pairs in total, fewer tree blocks than ranks) -- every rank answers alike, and the
 *     stage calls of section 5 serve that evaluation.
 *   icp_shard_loop_wait: the state after the launch; *evals = the evaluation ROUNDS it ran (an evaluation whose
 *     window missed is repeated once inside the launch with the widest windows and counts twice): what eval_base
 *     advances by; *finished, or *it = the evaluation the stage calls must serve
 *     next (its window missed, or the rotation left the range of the restated sin / cos).  ICP_HIP_ERROR: the launch
 *     gave up waiting for a peer (3 s) -- on EVERY rank: the rank whose wait ran out raises the abort word in all
 *     inboxes, so its peers leave at once and every host takes the same way out (nothing of the launch is used: the
 *     loop's state is the one it was started with; icp_reset_window_predictions, then the stage calls).
 * Ranks that share a device must launch on streams that really run side by side (they wait for each other): more
 * than four of them need GPU_MAX_HW_QUEUES raised before the HIP runtime starts. */
/* What an inbox is made of.  Memory that a PEER DEVICE writes while this device's kernels are polling it has to be
 * coherent between the two at every access, not only at kernel boundaries:
 *   ICP_INBOX_DEVICE  ordinary device memory (hipMalloc).  Right for ranks that share ONE device (virtual ranks of a
 *                     process; processes sharing a GPU through hipIpc): they share its L2.
 *   ICP_INBOX_FINE    fine-grained device memory (hipExtMallocWithFlags): peer access inside a process, hipIpc between
 *                     processes where the runtime exports such an allocation (icp_loop_inbox_ipc_handle fails otherwise).
 *   ICP_INBOX_HOST    pinned host memory in a POSIX shared-memory object: every process of the node maps it
 *                     (icp_loop_inbox_shm_name -> icp_loop_shm_open), registers it with its own device and gets a device
 *                     pointer.  Coherent by construction; the exchange crosses the host link instead of xGMI (a few
 *                     KB per evaluation).  The owner unlinks the name once every peer has opened it (_shm_unlink).
 * Which of them carries the loop between distinct devices is decided AT RUN TIME: icp_loop_transport_probe plays
 * ping-pong over the connected inboxes (every rank at once, bounded waits) and reports whether every token arrived; the
 * driver takes the first transport whose probe passes on every rank (icp_rust_amd/dist.py: connect_loop), else the
 * stage calls of section 5. */
#define ICP_INBOX_DEVICE 0
#define ICP_INBOX_FINE 1
#define ICP_INBOX_HOST 2
size_t icp_loop_inbox_bytes(void);
int icp_loop_inbox(icp_handle *h, int kind, void **d_inbox);
int icp_loop_inbox_ipc_handle(icp_handle *h, unsigned char out[64]);
int icp_loop_ipc_open(int device, const unsigned char handle[64], void **d_ptr);
int icp_loop_ipc_close(void *d_ptr);
int icp_loop_inbox_shm_name(icp_handle *h, char out[64]);
int icp_loop_inbox_shm_unlink(icp_handle *h);
int icp_loop_shm_open(int device, const char *name, void **d_ptr);
int icp_loop_shm_close(void *d_ptr);
int icp_shard_loop_connect(icp_handle *h, int rank, int world, void *const *inboxes);
int icp_loop_transport_probe(icp_handle *h, int rounds, int *ok);
/* every window prediction of the handle forgotten (what a rank does after a one-launch loop gave up: the stage calls
 * that serve from there on must see the same windows on every rank) */
int icp_reset_window_predictions(icp_handle *h);
int icp_shard_loop_launch_device(icp_handle *h, const double *d_a_xy_local, const double *d_b_xy_local, size_t n_total,
                                 unsigned launch_no, unsigned eval_base, int it0, uint32_t applied0, const icp_pose *Ti,
                                 double prev_error, int first_kind, int second_kind);
int icp_shard_loop_wait(icp_handle *h, icp_pose *Ti, double *prev_error, uint32_t *applied, int *it, int *finished,
                        uint32_t *evals);

/* ---- 5c. The PIPELINED sharded registration (round 6; csrc/pipe.hip, csrc/gn_win.hip: k_win_pick_shard) -------------
 * Replaces, for a rank of a sharded registration, the body of the reference's outer loop (src/lib.rs:113-127 / 156-170)
 * in the steady state of a registration -- the inner loop (src/lib.rs:59-84) applies exactly one update -- by the one-GPU
 * pipeline: search -> the two evaluations' first launches over the rank's own points -> their finishing workgroups, which
 * meet the other ranks' through the connected inboxes (section 5b: two exchanges per evaluation, no host, no collective)
 * and leave the next pose on the device for the search already enqueued behind them.  Three launches and one host wait
 * per outer iteration and rank.
 * icp_shard_pipe_run_device: collective -- every rank of the connection calls it at the same outer iteration *it_io with
 * the same pose *T_io (replicated state); it serves nothing until the stage calls or the loop launches have seeded the
 * window predictions of both kinds of evaluation (one outer iteration; a later call on the same handle starts from the
 * previous call's).  d_src_local: the rank's points (icp_shard_take_device out of the fold order,
 * icp_shard_prepare_source_device called on them).  On return *it_io / *T_io
 * are the iteration and pose the caller continues from; inner_iters[k] = 1 for every iteration k served (nullable);
 * d_idx_local receives the correspondences if the call's last iteration (max_iter - 1) was served.  *why = 0: served
 * through max_iter; 1: handed back (a window missed, an inner loop of no or several updates, a NaN: the caller's loop
 * serves iteration *it_io and may call again); 5: a wait for a peer ran out -- every rank reports it, the inboxes carry a
 * raised abort word, and the caller serves the rest through the stage calls.  The bits are those of the other two ways
 * to run a sharded evaluation, and of one GPU.  A rank may own up to 256 tree blocks (world x 2^20 points in total). */
int icp_shard_pipe_run_device(icp_handle *h, const double *d_src_local, size_t n_local, size_t n_total, int rank, int world,
                              icp_pose *T_io, size_t *it_io, size_t max_iter, uint32_t *inner_iters, uint32_t *d_idx_local,
                              int *why);

/* (STATUS: with every rank on ONE device -- "virtual ranks" -- this path runs in the test-suite; between DISTINCT
 * devices it has never run on hardware (no multi-GPU box was available to the build): experimental there.  Mixed lists
 * that repeat only some devices, e.g. {0, 0, 1, 1}, are refused.)
 * The same from ONE host process over the GPUs of a node (SURVEY.md 8(b) sketch: device_ids /
 * n_devices): rank r is a handle on device_ids[r], the target cloud is replicated, the source cloud
 * sharded by reduction-tree block, and the two exchanges are peer reads over xGMI behind flags (no
 * collective library: there is no ICP_RCCL_ERROR).  The result equals icp_estimate's on one GPU bit
 * for bit.  device_ids may repeat a device ("virtual ranks": how the N-rank path is tested on a
 * one-GPU box; ranks on one device share a stream).  icp_multi_counters: out[0] evaluations that ran
 * sharded, out[1] evaluations that fell back to gathered pairs. */
typedef struct icp_multi icp_multi;
int icp_create_multi(icp_multi **out, int dim, const double *dst, size_t m, const int *device_ids, int n_devices);
int icp_multi_estimate(icp_multi *M, const double *src, size_t n, const icp_pose *init, size_t max_iter,
                       icp_pose *out, uint32_t *last_idx, uint32_t *inner_iters);
/* EXTENSION (section 6 across the ranks; BASELINE configs[4] on several GPUs): every rank appends the same k points,
 * moved by T (NULL: as they are), to its replica of the target cloud and rebuilds its search grid; afterwards the
 * object answers like a fresh icp_create_multi on the concatenated cloud, in the sense of section 6 (correspondences
 * always; pose bits for source clouds that keep the caller's fold order). */
int icp_multi_append_targets(icp_multi *M, const double *pts, size_t k, const icp_pose *T);
size_t icp_multi_target_count(const icp_multi *M);
void icp_destroy_multi(icp_multi *M);


/* The N-term sums (jtj, jtr, Huber error) are added in a fixed, run-to-run
 * deterministic tree; this reports its geometry for n points so a checker can
 * reproduce the exact association order (DESIGN.md "GN reduction order"). */
void icp_reduce_geometry(size_t n, int *blocks, int *threads);
/* ... over the source points in FOLD ORDER.  The reference folds over the caller's order
 * (src/lib.rs:240-255, a left fold); the sum is the same up to rounding, so any fixed order meets the
 * 1e-5 pose bar.  When icp_estimate[_device] takes a cell-sorted snapshot of the source cloud (grid
 * engine, MORE THAN 65 536 source points -- the largest cloud searched with four lanes per query; the threshold is a
 * constant of the product library: the development build libicp_mi355x_exp.so (`make experiments`) can move it with
 * ICP_NN_COOP_MAX_N, which moves result bits with it) the fold order IS the snapshot order -- ascending
 * (target-grid cell of init * src[i], i), a stable sort, hence a pure function of the inputs AND OF THE HANDLE'S GRID
 * (box and cell size: what icp_create chose for the target cloud, kept by incremental appends, section 6) -- so that
 * the search can store its pairs with full-line writes and nothing is ever scattered back; otherwise (smaller clouds,
 * the sweep engine, the stage calls) it is the caller's order.
 * icp_last_fold_order reports it for the last estimate call on `h` (host buffers of n words, either may be
 * NULL): perm[k] = caller's index of the k-th folded point, cell[k] = its sort key (all 0 for the identity).
 * A checker reproduces the device sums by folding the pairs of src[perm[0]], src[perm[1]], ... in the
 * tree of icp_reduce_geometry. */
int icp_last_fold_order(icp_handle *h, size_t n, uint32_t *perm, uint32_t *cell);
/* The fold order of an estimate call that starts at pose T, applied (device buffers): d_sorted[k] = d_src[perm[k]],
 * d_perm nullable.  For hosts that drive the stage calls of sections 4 and 5 themselves and want the bits of
 * icp_estimate_device: sort first, then treat the sorted cloud as the source (sorting it again is the identity). */
int icp_sort_source_device(icp_handle *h, const double *d_src, size_t n, const icp_pose *T, double *d_sorted,
                           uint32_t *d_perm);




/* The reference builds a new Icp per frame (examples/scan2d.rs:87, scan3d.rs:130), so icp_destroy
 * keeps the device buffers, streams and pinned memory of up to two handles per process for the
 * next icp_create on the same device (a create then costs its kernels, not its allocations).
 * icp_trim_pool releases them. */
void icp_trim_pool(void);

/* ================================================================================
 * 6. EXTENSION (not in the reference): a target cloud that grows -- scan-to-map
 * ==============================================================================
 * BASELINE.json configs[4] registers each scan against a map that the
 * registered scans are appended to.  The reference has no such thing: its Icp borrows a fixed
 * `dst` (src/lib.rs:97-102, 139-144).  What the reference does define is the registration of a
 * scan against ANY target cloud, so a map handle is an ordinary handle whose target cloud can
 * be extended; after an append every CORRESPONDENCE is that of a fresh icp_create on the
 * concatenated cloud (target indices = position in the concatenation), and so is every pose, bit for bit, as long as
 * the source cloud keeps the caller's fold order (up to 65 536 points, section 9a).  Larger source clouds fold
 * their sums in the order of the handle's grid cells; an incrementally appended handle keeps the grid of its last full
 * build where a fresh handle derives a new one, so their poses agree to the rounding of a re-ordered sum (~1e-12
 * relative; the bar is 1e-5) -- each equals the oracle evaluated in ITS fold order (icp_last_fold_order) bit for bit.  Point-to-plane
 * residuals, also named by configs[4], have no definition in the reference (no normals
 * anywhere in src/): section 7 builds them as a second labelled extension.
 *
 * icp_append_targets[_device]: append k points (AoS, the handle's dim) to the target cloud;
 * with T != NULL the points are first moved by T exactly as Transform::transform does
 * (transform.rs:22-24: x' = (r00 x + r01 y) + tx, y' = (r10 x + r11 y) + ty, z kept; no FMA),
 * i.e. "append the scan at its registered pose".  A handle made by icp_create_device stops
 * borrowing the caller's buffer at its first append (the cloud moves into storage the handle
 * owns, grown geometrically).  The search structures are rebuilt on the device.
 * icp_reserve_targets sizes that storage ahead of time; icp_target_count reports m. */
int icp_append_targets(icp_handle *h, const double *pts, size_t k, const icp_pose *T);
int icp_append_targets_device(icp_handle *h, const double *d_pts, size_t k, const icp_pose *T);
int icp_reserve_targets(icp_handle *h, size_t capacity);
size_t icp_target_count(const icp_handle *h);
/* copy target points [first, first + k) back to the host (AoS), e.g. to save the map */
int icp_read_targets(icp_handle *h, size_t first, size_t k, double *out);

/* ================================================================================
 * 7. EXTENSION (not in the reference): point-to-plane residuals
 * ==============================================================================
 * The second thing BASELINE.json configs[4] names.  tier4/icp_rust is point-to-point only (no normal
 * anywhere in src/; src/lib.rs:218-261), so there is no reference behaviour to match and no parity
 * claim: the definition lives in icp_rust_amd/csrc/p2plane.hip, its CPU restatement (the checker of
 * tests/test_p2plane.py) in oracle/icp_oracle.c (orc_p2pl_*).  Kept from the reference: the exact 3-D
 * nearest neighbour, the SE(2) pose on xy with z carried through, the Huber / MAD Gauss-Newton loop
 * with its constants and break tests, the 3x3 solve, Transform::new.  3-D handles only.
 *
 * icp_compute_target_normals: unit normal of every target = smallest-eigenvalue eigenvector of the
 * covariance of its k nearest targets (3 <= k <= 16, itself included), searched through the handle's
 * grid; after an append call it again, or icp_update_target_normals: only the appended targets get a
 * normal (from the cloud as it is now), the older ones keep theirs ("normals at insertion time": what a
 * map that grows frame by frame uses; same k as before).  icp_estimate_point_to_plane*: Icp3d::estimate
 * with the scalar residual n_q . (T p - q) in place of the two-row point-to-point residual. */
int icp_compute_target_normals(icp_handle *h, int k);
int icp_update_target_normals(icp_handle *h, int k);
int icp_read_target_normals(icp_handle *h, size_t first, size_t count, double *out_xyz);
int icp_estimate_point_to_plane(icp_handle *h, const double *src, size_t n, const icp_pose *init,
                                size_t max_iter, icp_pose *out, uint32_t *last_idx, uint32_t *inner_iters);
int icp_estimate_point_to_plane_device(icp_handle *h, const double *d_src, size_t n, const icp_pose *init,
                                       size_t max_iter, icp_pose *out, uint32_t *d_last_idx,
                                       uint32_t *inner_iters);
/* ... across the ranks of an icp_multi (round 4; BASELINE configs[4] on several GPUs): the normals belong to the replicated
 * target cloud, so every rank computes / updates the same ones; a registration shards the SEARCH (contiguous slices of
 * the source cloud), every rank receives every slice's indices and runs the same inner loop on the whole cloud
 * (SURVEY.md 8(e) option 1).  Result: one handle's icp_estimate_point_to_plane, bit for bit. */
int icp_multi_compute_target_normals(icp_multi *M, int k);
int icp_multi_update_target_normals(icp_multi *M, int k);
int icp_multi_estimate_point_to_plane(icp_multi *M, const double *src, size_t n, const icp_pose *init, size_t max_iter,
                                      icp_pose *out, uint32_t *last_idx, uint32_t *inner_iters);

#ifdef __cplusplus
}
#endif
#endif /* ICP_MI355X_H */
