// Shared declarations of the device side: handle layout, workspace, kernel launchers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <utility>
#include <vector>

#include "pose.hpp"

namespace icp {

// The product library reads FIVE environment variables -- ICP_NO_POOL (icp_destroy frees instead of pooling),
// ICP_NO_GN_LOOP (inner loops stepped from the host only), ICP_NO_SPECULATION (no bet on the next pose),
// ICP_GN_NO_REFINE (no refined windows beyond 4M pairs), ICP_MULTI_DEBUG (icp_multi diagnostics on stderr).  Every
// A/B switch and tuning knob of the development rounds exists only in a build with
// -DICP_EXPERIMENTS (`make experiments` -> libicp_mi355x_exp.so), where exp_env is getenv; here it answers "unset".
#ifdef ICP_EXPERIMENTS
inline const char *exp_env(const char *name) { return getenv(name); }
#else
inline const char *exp_env(const char *) { return nullptr; }
#endif

// roctx ranges around the three device stages of an outer iteration (search, evaluation / inner
// loop, solve), for `rocprofv3 --marker-trace`.  The marker library is looked up at run time (librocprofiler-sdk-roctx
// or libroctx64): the product links nothing but libamdhip64, and without the library the ranges cost one branch.
struct Range {
  explicit Range(const char *name);
  ~Range();
  Range(const Range &) = delete;
  Range &operator=(const Range &) = delete;
};

constexpr int kReduceThreads = 512;   // threads per block of the GN reduction tree
constexpr int kReduceMaxBlocks = 256;  // blocks of the tree up to 2^20 points (one per CU), and of ONE rank's share beyond
// Round 6: beyond 2^20 points the tree GROWS with the cloud -- a block per 4 096 points (eight per thread, as at 2^20),
// up to kTreeMaxBlocks (2^23 points: eight ranks' 1M each; larger clouds fold more per thread again) -- instead of keeping 256 blocks whose threads fold more and more points: a rank of an N-GPU
// registration of N x 1M points then owns 256 blocks and evaluates on all its CUs, with the launches one GPU runs on 1M
// points.  The second stage folds the block sums as before: thread t of one 512-thread block takes
// rows t, t + 512, ...  Bits of clouds beyond 2^20 points changed once with this.
constexpr int kTreeMaxBlocks = 2048;
constexpr int kShardMaxWorld = 16;  // ranks of one sharded evaluation (a node has 8 GPUs)
constexpr int kNAcc = 13;             // what a weighted evaluation hands the host: jtj[9], jtr[3], huber error
// Round 3: the device folds the weighted normal equations PER DIMENSION j, WITHOUT the factor g_j = 1 / sigma_j, and
// only the upper triangle of J^T W J:
//   S_j[u(p,q)] = sum_i (w_ij J_ij[p]) J_ij[q]   (p <= q; u = 0..5 for 00 01 02 11 12 22)
//   S_j[6 + k]  = sum_i (w_ij J_ij[k]) r_ij
// and applies g to the folded totals: jtj[p][q] = g_x S_x[u] + g_y S_y[u], mirrored (combine_sum).  The reference
// multiplies every term by w g first and evaluates all nine products (src/lib.rs:246-254); the two differ by rounding
// only.  Without sigma in the terms the sums no longer have to wait for the four exact medians:
// they ride in the pass that computes the residuals (gn_win.hip: k_win_hist_sums).
constexpr int kNSum = 19;             // S_x[9] | S_y[9] | huber error
constexpr int kSelProblems = 4;       // {x, y} x {lower, upper middle order statistic}
constexpr int kSelBins = 4096;        // 12-bit radix digits
constexpr int kSelPasses = 6;         // 12+12+12+12+12+4 bits
constexpr int kSelRoles = 6;          // fast path: histogram buffers {median, MAD} x {digit 0, 1, 2}
constexpr int kSelCap = 1024;         // fast path: candidates kept per problem after two passes
// window path (gn_win.hip): one piecewise-linear histogram per dimension instead of radix digits
constexpr int kWinFine = 512;         // bins of each of the three fine windows (median, median -+ MAD)
constexpr int kWinCoarse = 255;       // bins of each of the two stretches between them
constexpr int kWinBins = 2 + 3 * kWinFine + 2 * kWinCoarse;  // + everything below / above
constexpr int kWinCapMed = 1024;      // candidates around the median, per dimension
constexpr int kWinCapRing = 4096;     // candidates around median -+ MAD, per dimension
constexpr int kWinBlocks = 256;       // workgroups of the streaming launches (one per CU)
constexpr int kWinBlkMed = 32;        // candidates one workgroup can stage, per dimension
constexpr int kWinBlkRing = 96;
// filed candidates (gn_win.hip: k_win_hist_sums_bkt / k_win_pick): the pass that counts the residuals also files every
// residual of a FINE bin -- sorted by bin, in the workgroup's own segment, with a directory of where each fine bin's
// members start -- so the candidates of whatever bins the counts resolve to are already lying there: no second pass
// over the points.  Nothing is shared between workgroups (returning atomics on shared counters serialise per counter:
// 6.5 us for 256 workgroups, measured).
constexpr int kBktStage = 2048;       // fine-window members one workgroup can stage (LDS) and file (its segment)
constexpr int kBktFine = 2 * 3 * kWinFine;  // fine bins of both dimensions, in directory order: dimension, window, bin
constexpr int kBktDir = kBktFine + 8;       // directory entries per workgroup (u16 offsets; [kBktFine] = the total), padded

inline void reduce_geometry(size_t n, int *blocks, int *threads) {
  size_t b = (n + kReduceThreads - 1) / kReduceThreads;
  if (b < 1) b = 1;
  if (b > (size_t)kReduceMaxBlocks) b = kReduceMaxBlocks;
  if (n > ((size_t)kReduceMaxBlocks * kReduceThreads * 8)) {  // beyond 2^20 points: a block per 4 096
    b = (n + (size_t)kReduceThreads * 8 - 1) / ((size_t)kReduceThreads * 8);
    if (b > (size_t)kTreeMaxBlocks) b = kTreeMaxBlocks;
  }
  *blocks = (int)b;
  *threads = kReduceThreads;
}

// One exact order-statistic search (radix select state), device resident.
struct SelState {
  unsigned long long prefix;  // key bits fixed so far (high bits)
  unsigned long long rank;    // rank of the wanted element inside the prefix group
  int alias;                  // >= 0: shares the histogram of that problem (same prefix)
  int pad;
};

// Device-resident scalars of one inner iteration.
struct GnScalars {
  double median[2];
  double sigma[2];
  int nan_flag;
  int overflow;  // fast selection path gave up (too many candidates): redo with the radix path
};

// Arrival tickets of one kernel role: 16 shard counters + one top counter, each on a
// 128-B line of its own (a single word saturates at ~88 returning atomics per us; ~500
// workgroups finishing together would queue for ~6 us on it).
struct TicketSet {
  unsigned shard[16][32];
  unsigned top[32];
};

// Tickets and candidate counters of the fast selection path (all zero between launches).
struct SelCtl {
  TicketSet t[3];
  unsigned cand_cnt_pull[2][kSelProblems];  // gn_pull.hip: candidates per stage (median, MAD)
  unsigned pad[20];
};

// The bins of one dimension: (-inf, x[0]) | fine | coarse | fine | coarse | fine | [x[5], inf),
// the fine windows centred on the predicted median - MAD, median, median + MAD.
struct WinDim {
  double x[6];
  double sf, sc;  // bins per unit inside the fine windows / the coarse stretches
};
struct WinParams {
  WinDim d[2];
};

// What k_win_compact resolves from the histograms for k_win_accumulate (per dimension).
struct WinState {
  unsigned fail;           // the window missed an order statistic: evaluate again with gn_pull.hip
  unsigned stage_overflow; // bucketed candidates: a workgroup met more fine-window members than it can stage (k_win_pick clears it)
  unsigned med_base[2];    // points in bins below the median bins
  unsigned med_cnt[2];     // points in the median bins (= candidates the compaction delivers)
  unsigned ring_inner[2];  // points surely closer to the median than the MAD
  unsigned ring_cnt[2];    // points that may be at MAD distance (= candidates)
  double med_lo[2], med_hi[2];    // value range of the median candidates
  double ring_lo[2], ring_hi[2];  // distance range that holds the MAD
  unsigned list_cnt[4][32];       // appended so far {med x, med y, ring x, ring y}, a 128-B line each
};

// What the last kernel of an inner iteration hands to the host (pinned, mapped).
struct GnResult {
  double acc[kNAcc + 1];  // jtj[9], jtr[3], huber error, (plain) error
  double sigma[2];
  double median[2];       // gn_pull.hip / gn_win.hip only: centre of the next evaluation's window
  int nan_flag;
  int overflow;  // 1: too many candidates (radix path needed), 2: the window missed (gn_pull.hip needed)
  unsigned seq;  // written last, system scope: the host polls it instead of waiting for the stream
  unsigned pad;
  // sharded evaluations: how many ranks answered {OK, RETRY_REPLICATED, NONE, anything else} in the hist stage
  // (the status words behind the histograms, summed over the ranks with them; shard.hip:k_shard_fold)
  unsigned status[4];
  // k_win_finish with an AheadPose to fill (icp_estimate_device's run-ahead search): the pose it derived and whether
  // it is usable -- the host compares it bit for bit with the pose it computes itself from acc[]
  Pose next_pose;
  int next_valid;
  int pad2;
};

// The pose of a search that was enqueued BEFORE the host knew it: the finishing launch of an outer iteration's first
// evaluation solves the update on the device and leaves "update x outer pose" here; the search behind it on the same
// stream reads it (valid == 0: every wave leaves at once).  Results never depend on it: the host derives the same pose
// from the evaluation's sums and uses the pairs only if the two agree bit for bit.
struct AheadPose {
  Pose T;
  int valid;
  int pad;
};

// Scratch of ONE Gauss-Newton evaluation in flight.  A handle has two of them, one per stream of
// icp_estimate_device (an evaluation that decides a speculated pose and the speculated next
// iteration's first evaluation are in flight together); `Workspace` derives from the active one,
// so the launchers simply see `w.d_rx` etc., and switching the stream swaps it with `alt`.
struct GnCtx {
  double *d_rx = nullptr;  // residual x, cap_n
  double *d_ry = nullptr;  // residual y, cap_n
  // selection + reduction scratch (fixed size)
  uint32_t *d_hist = nullptr;   // kSelRoles x kSelProblems x kSelBins (role 0 also serves the radix path)
  unsigned long long *d_cand = nullptr;  // 2 stages x kSelProblems x kSelCap candidate keys
  SelCtl *d_ctl = nullptr;
  SelState *d_sel = nullptr;    // 2 x kSelProblems (gn_pull.hip ping-pongs between the halves)
  GnScalars *d_scal = nullptr;
  double *d_partials = nullptr; // kTreeMaxBlocks x (kNSum+1)
  GnResult *h_res = nullptr;    // pinned coherent host memory, written by the last workgroup
  unsigned seq = 0;             // sequence number of the last fast evaluation launched
  bool gn_dirty = true;         // selection scratch is not in its all-zero rest state: k_sel_init first
  // window path (gn_win.hip)
  uint32_t *d_whist = nullptr;  // 2 x kWinBins, zero between evaluations
  WinState *d_wstate = nullptr;
  double *d_wmed = nullptr;     // 2 x kWinCapMed residuals
  double *d_wring = nullptr;    // 2 x kWinCapRing residuals
  double *d_bkt = nullptr;      // kReduceMaxBlocks segments of kBktStage residuals: a workgroup's fine-window members, by bin
  unsigned short *d_bkt_dir = nullptr;  // kReduceMaxBlocks x kBktDir: where each fine bin's members start in the segment
  bool bkt_pair_launched = false;  // this context's evaluation is in flight on the search stream (launch_bkt_pair): only its result is awaited
};

struct Workspace : GnCtx {
  GnCtx alt;               // the other stream's evaluation scratch
  void swap_ctx() { std::swap(static_cast<GnCtx &>(*this), alt); }
  size_t cap_n = 0;        // points the per-point buffers hold
  double *d_src = nullptr; // staged source cloud (host API), cap_n x dim
  double *d_a = nullptr;   // transformed source xy, cap_n x 2
  double *d_b = nullptr;   // matched target xy, cap_n x 2
  double *d_a2 = nullptr;  // second pair buffers: the speculative search of the next outer iteration
  double *d_b2 = nullptr;
  double *d_a3 = nullptr;  // third: the run-ahead search of the iteration after that
  double *d_b3 = nullptr;
  AheadPose *d_ahead = nullptr;   // device: the run-ahead search's pose (k_win_finish writes, k_nn_grid_warm_coop reads)
  bool ahead_on = false;          // the next two-launch evaluation fills d_ahead for ...
  Pose ahead_outer;               // ... this outer pose
  bool ahead_seen_valid = false;  // what the last pre-launched evaluation reported (GnResult::next_pose / next_valid)
  Pose ahead_seen_pose;
  unsigned long long ahead_hits = 0, ahead_misses = 0;
  double dbg_wait_pre_us = 0., dbg_wait_other_us = 0.;  // experiments build: where the host of icp_estimate_device waits
  unsigned long long spec_hits = 0, spec_misses = 0, pre_evals = 0;
  uint32_t last_inner = 0xffffffffu;  // updates the inner loop applied in the last outer iteration of the previous call
  hipStream_t spec_stream = nullptr;  // later evaluations of an inner loop run beside the speculative search
  bool search_beside_eval = false;    // this outer iteration bets on a speculative search: its deciding evaluation shares the CUs with it
  uint32_t *d_idx = nullptr;
  uint32_t *d_idx_slot = nullptr;  // the last search's indices in slot order (icp_estimate_device, QuerySort::slot_order)
  // refined windows (n > 4M, gn_win.hip): a strided sample of the pairs and a host copy of the histograms
  double *d_sa = nullptr, *d_sb = nullptr;
  uint32_t *h_whist = nullptr;
  void *h_tiny = nullptr;  // pinned result block of the one-launch registration of small clouds (gn_fast.hip)
  unsigned long long tiny_calls = 0, tiny_evals = 0, tiny_sorted = 0;
  double *d_rlist = nullptr;        // 2 x kRefineListCap: the residuals inside the second pass' fine windows
  unsigned *d_rlist_len = nullptr;  // [2]
  unsigned long long refine_tried = 0, refine_missed = 0;
  // brute-force NN partial minima when the target range is split over blockIdx.y
  size_t cap_part = 0;
  double *d_part_d = nullptr;
  uint32_t *d_part_i = nullptr;
  // prediction for the window path (host state, shared by both contexts)
  bool win_valid = false;       // median/sigma of the previous evaluation are known
  bool win_wide = false;        // the last window missed: use wider fine windows until it settles
  double win_med[2] = {0., 0.}, win_sigma[2] = {0., 0.};
  // An inner loop alternates between two populations of residuals: the first evaluation on new
  // correspondences (kind 0) and the evaluation after the first update (kind 1).  Their medians
  // differ by 0.1-0.3 sigma while consecutive evaluations of the SAME kind differ by ~0.01 sigma
  // (measured on 28k-point frames), so each kind is predicted from its own previous evaluation;
  // the fields above (the most recent evaluation of any kind) serve the later evaluations of a
  // loop and whatever has no history of its own yet.
  struct WinPred {
    bool valid = false, wide = false;
    double med[2] = {0., 0.}, sigma[2] = {0., 0.};
  };
  // kinds 3 and 4 = the first and the second evaluation of a CALL's first outer iteration: a new call
  // (the next frame, or the same cloud again) starts from a pose the previous call's last evaluations
  // say nothing about, but it usually resembles the start of the previous call.  Their statistics also
  // seed kinds 0 and 1 for the call's second iteration.
  WinPred win_kind[5];
  // HINTS from the previous owner of a pooled handle (examples/scan3d.rs creates an Icp3d per frame, and the next frame
  // resembles the last): its per-kind predictions and how its last inner loop ended.  Adopted once, by the one-GPU
  // evaluation of that kind (api.hip: wgn_step) -- a prediction can cost a repeated evaluation, never change a result
  // -- and never by the sharded stages, whose branch decisions may only depend on replicated state.
  WinPred hint_kind[5];
  uint32_t hint_last_inner = 0xffffffffu;
  static bool kind_has_slot(int kind) { return kind == 0 || kind == 1 || kind == 3 || kind == 4; }
  unsigned long long win_tried = 0, win_missed = 0, short_evals = 0, radix_evals = 0;
  // bucketed candidates (gn_win.hip): evaluations served that way, how many found their buckets too small, and for how
  // many more window evaluations this handle keeps to the second pass over the points after such a miss
  unsigned long long bkt_evals = 0, bkt_misses = 0;
  unsigned long long fixed_point_skips = 0;  // outer iterations not run because the pose had stopped moving (icp_estimate_device)
  unsigned bkt_off = 0;
  // one-launch inner loop (gn_loop.hip): control block, the two parity histograms / block-sum sets, the pinned result
  void *d_loop_ctl = nullptr, *h_loop_res = nullptr;
  uint32_t *d_loop_hist = nullptr;
  double *d_loop_part = nullptr;
  unsigned loop_seq = 0;
  bool loop_off = false;  // a launch was not resident (its grid barrier timed out): this handle steps from the host
  unsigned long long loop_launches = 0, loop_evals = 0, loop_handbacks = 0, loop_timeouts = 0;
  // ... over the ranks of a sharded registration (gn_loop.hpp: LoopInbox): this rank's inbox, every rank's as mapped
  // here, and the launch in flight (api.hip: icp_shard_loop_launch_device / icp_shard_loop_wait)
  void *d_loop_inbox = nullptr;
  int loop_inbox_kind = 0;           // ICP_INBOX_DEVICE / _FINE / _HOST (include/icp_mi355x.h section 5b)
  void *loop_inbox_host = nullptr;   // _HOST: the mapping of the shared-memory object behind d_loop_inbox
  char loop_shm_name[64] = {0};      // ... and its name (unlinked once every peer has opened it)
  unsigned loop_probe_gen = 0;       // tokens of the transport probes so far
  unsigned loop_off_calls = 0;       // launches left to skip after one that was not resident (loop_off decays)
  void *loop_peers[kShardMaxWorld] = {};
  int loop_rank = -1, loop_world = 0;
  void *loop_plan = nullptr;  // LoopPlan of the launch in flight (api.hip)
  // the pipelined sharded evaluation (pipe.hip): generations of its exchanges so far (the same on every rank of the
  // connection), iterations it served / handed back / gave up on, iterations it keeps away after a miss of its files
  unsigned pipe_gen = 0;
  unsigned long long pipe_iters = 0, pipe_handbacks = 0, pipe_gave_up = 0;
  unsigned pipe_off = 0;
};

// ---- uniform grid over the target cloud (nn_grid.hip) ------------------------------
struct GridParams {
  double lo[3];
  double hi[3];
  double h[3], inv_h[3];  // cell size per axis (x may be finer: rows are contiguous along x)
  int n[3];
  int fx;        // x cells per y/z cell size: the cold search grows its block by fx cells along x per ring
  double scale;  // coordinate magnitude used for the rounding margin of the pruning bounds
  float ext;     // largest extent of the bounding box + one cell: magnitude of every grid-relative coordinate
  int f32_ok;    // the grid-relative geometry is representable in f32 with room to spare (k_nn_grid_warm)
  // the f32 geometry's constants, rounded once on the host ((float) of the doubles above: the conversions the warm
  // kernels used to repeat in every wave)
  float hf[3], ihf[3], nm1f[3];  // cell size, its inverse, n - 1
};

struct GridPoint {  // one target, cell-sorted: a 16-B pre-filter record = one load per candidate
  float x, y, z;   // fl32(coordinate - grid lo); exact f64 coordinates are read from dst by idx
  uint32_t idx;    // original index in dst
};

struct Grid {
  bool built = false;
  GridParams p;
  uint32_t ncell = 0;
  uint32_t *d_start = nullptr;  // ncell + 1 cell offsets into d_pts
  GridPoint *d_pts = nullptr;   // m targets sorted by cell, then kGridPad sentinel records (+inf: never pass a screen)
  // capacities (elements) and build temporaries: kept, so that a pooled handle rebuilds without allocating
  size_t cap_start = 0, cap_pts = 0;
  uint32_t *t_cell_of = nullptr, *t_cnt = nullptr, *t_btot = nullptr;
  size_t cap_tcell = 0, cap_tcnt = 0, cap_tbtot = 0;
  double *t_part = nullptr;     // bounding-box partials
  // incremental append (append_grid): the cell of the record at every sorted position, and a second set of the
  // sorted arrays (records move out of place by their cell's shift, then the sets swap)
  uint32_t *d_rcell = nullptr, *d_rcell2 = nullptr, *d_start2 = nullptr, *t_shift = nullptr, *t_cell_new = nullptr;
  GridPoint *d_pts2 = nullptr;
  size_t cap_rcell = 0, cap_rcell2 = 0, cap_start2 = 0, cap_pts2 = 0, cap_shift = 0, cap_cell_new = 0;
  size_t m_full = 0;            // targets at the last full build (the cell size was chosen for that many)
  bool rcell_valid = false;     // d_rcell describes the current records
  unsigned long long appends_moved = 0, appends_rebuilt = 0;  // observability (icp_grid_append_counters)
  uint32_t *d_flag = nullptr;   // one word: a new point outside the grid's box (or not finite)
};

constexpr int kShardStatusWords = 4;  // behind the 2 x kWinBins histogram words of a sharded evaluation
constexpr unsigned kGridPad = 8;  // records past the last target that a quad-aligned read may touch

// sharded evaluation (shard.hip): the head of the block a rank hands to the others, and one pointer per rank
struct ShardCandHeader {
  unsigned cnt[4];  // appended {med x, med y, ring x, ring y}
  unsigned fail;    // this rank's compaction missed (identical on every rank: same histogram)
  unsigned pad[11];
};
struct ShardPtrs {
  const unsigned char *p[kShardMaxWorld];
};

struct PrevMatch {  // a query's previous nearest neighbour, stored per sorted slot (coalesced)
  double x, y, z;
  uint32_t idx, pad;
};

// cell-sorted copy of a source cloud (prepare_queries): locality for the grid search.  The order is
// DETERMINISTIC: ascending (target-grid cell of T0 * src[i], i) -- a stable sort -- so that it can serve
// as the order in which icp_estimate_device folds its sums (`slot_order`).
struct QuerySort {
  bool valid = false;
  // icp_estimate_device: the searches of this call emit a / b / idx in SLOT order (coalesced stores, no
  // scatter through `perm`) and the Gauss-Newton evaluations fold the pairs in that order
  bool slot_order = false;
  bool identity = false;       // the snapshot keeps the caller's order (no sort: clouds of up to ICP_NN_COOP_MAX_N points)
  bool presorted = false;      // one-shot hint: the next snapshot's cloud is already in cell order (a rank's slice of a sorted cloud)
  bool sort_only = false;      // one-shot: the next prepare_queries only sorts (d_perm; no sorted copy, no snapshot): icp_shard_sort_take_device
  size_t fold_n = 0;           // > 0: d_perm / d_cell hold the fold order of the last estimate call on fold_n points
  const double *src = nullptr;  // the device buffer this snapshot was taken from
  size_t n = 0, cap = 0;
  uint32_t *d_cell_of = nullptr;  // cell of every source point, original order (sort keys in)
  uint32_t *d_perm = nullptr;     // slot -> original index
  void *d_tmp = nullptr;          // the radix sort's temporary storage
  size_t cap_tmp = 0;
  bool have_prev = false;      // d_prev holds the matches of an earlier search of this snapshot
  PrevMatch *d_prev = nullptr; // per sorted slot: the last match (idx = ~0u: none)
  double *d_sorted = nullptr;
  // certified matches (nn_grid.hip: k_nn_cert): how far the searches of this snapshot have moved the queries so far
  // (per unit of |s_xy| and flat; upper bounds, accumulated in launch order), the previous search's pose, the work
  // lists of the queries whose certificate failed and their counters (two sets, alternating between searches)
  bool have_certs = false, have_pose = false, have_pose_before = false;
  double decay_r = 0., decay_t = 0.;
  Pose last_pose;
  uint32_t *d_cert_lists = nullptr;
  unsigned *d_cert_ctr = nullptr, *last_cert_ctr = nullptr;
  unsigned cert_seq = 0;
  unsigned long long cert_searches = 0;
};

// stable LSD radix sort of the indices 0 .. n-1 by cell (qsort.hip): perm_out[k] = index of the k-th point in (cell, index) order
hipError_t stable_sort_cells(const uint32_t *keys_in, uint32_t *perm_out, unsigned n, unsigned bits, void *&tmp, size_t &cap_tmp,
                             hipStream_t s);

}  // namespace icp

struct icp_handle {
  int dim = 0;
  size_t m = 0;
  int device = 0;
  int nn_mode = ICP_NN_AUTO;
  bool single_launch = true;  // small clouds: the whole estimate in one launch (icp_set_single_launch)
  bool fixed_point_exit = true;  // icp_estimate_device leaves out the iterations behind a fixed point (icp_set_fixed_point_exit)
  bool owns_dst = false;
  const double *d_dst = nullptr; // AoS m x dim (the owned copy below, or borrowed)
  double *d_dst_own = nullptr;   // buffer for a host-supplied target cloud
  size_t cap_dst_own = 0, cap_soa = 0, cap_f32 = 0;  // capacities in elements (buffers survive in the handle pool)
  double *d_dst_soa = nullptr;   // x[m_pad] | y[m_pad] | z[m_pad], padded with +inf
  float *d_dst_f32 = nullptr;    // fl32(p - bbox lo), same layout: the sweep's f32 screen
  size_t m_pad = 0;
  bool brute_valid = false;      // d_dst_soa describes the current target cloud (else launch_nn_brute rebuilds it)
  bool screen_valid = false;     // d_dst_f32 too (needs the grid's bounding box: finite targets only)
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  icp::Workspace ws;
  icp::Grid grid;
  icp::QuerySort qsort;
  // sharded evaluation in flight (shard.hip): what eval_hist decided, for the stages that follow
  struct ShardEval {
    icp::WinParams P;
    int kind = 2, rank = 0, world = 1, b0 = 0, b1 = 0, blocks = 0;
    size_t n_local = 0, n_total = 0;
    const double *d_a = nullptr;
    icp::Pose T;
    double *d_ordered = nullptr;  // kTreeMaxBlocks x (kNSum + 1): the block sums of all ranks in block order
    bool active = false;
    icp::WinParams P2;            // the window refined from a missed attempt's global counts
    bool refined_ready = false, attempt_refined = false;
  } shard;
  // EXTENSION (p2plane.hip): unit normals of the target points (m x 3): those of targets [0, normals_m) exist;
  // usable while normals_m == m (an append leaves the new targets without one: icp_update_target_normals)
  double *d_normals = nullptr;
  size_t cap_normals = 0, normals_m = 0;
  int normals_k = 0;
  void *d_plane_pairs = nullptr;  // per-pair constants of a point-to-plane inner loop
  double *d_plane_fa = nullptr, *d_plane_fb = nullptr;
  size_t cap_plane = 0;
  // live kernel timing (icp_profile_*): event pairs around the NN search kernel
  int profile = 0;         // 0: off; k: event pairs around every k-th search launch
  unsigned prof_seen = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_free;  // event pairs read out, kept for the next samples
};

namespace icp {

// grow-only device buffer: reallocates (with 1/8 headroom) when `need` elements exceed `cap`
template <typename Tp>
inline hipError_t reserve(Tp *&p, size_t &cap, size_t need) {
  if (need <= cap && p) return hipSuccess;
  if (p) {
    (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  const size_t want = need + need / 8 + 1;
  const hipError_t e = hipMalloc(&p, want * sizeof(Tp));
  if (e == hipSuccess) cap = want;
  return e;
}

// ---- launchers (each enqueues on h->stream and returns the HIP error) ---------------
hipError_t ensure_workspace(icp_handle *h, size_t n, bool need_src);
hipError_t build_target_soa(icp_handle *h);
hipError_t build_target_screen(icp_handle *h);  // after build_grid

// transform (optional) + brute-force exact NN + gather of the matched xy pairs
hipError_t launch_nn_brute(icp_handle *h, const double *d_src, size_t n, const Pose *T,
                           double *d_a, double *d_b, uint32_t *d_idx);

hipError_t launch_materialize(icp_handle *h, const double *d_src, size_t n, const Pose &T, const uint32_t *d_idx,
                              double *d_a, double *d_b);
// exact uniform-grid NN: same outputs, same results as launch_nn_brute
hipError_t build_grid(icp_handle *h);
// the grid after k targets were appended behind the first m_old: moves the sorted records by their cells' shifts and
// inserts the new ones (*done = false: not applicable -- no grid, a point outside its box, the cloud outgrew its cell
// size -- and the caller rebuilds)
hipError_t append_grid(icp_handle *h, size_t m_old, size_t k, bool *done);
hipError_t prepare_queries(icp_handle *h, const double *d_src, size_t n, const Pose &T);
long grid_coop_max();
hipError_t launch_unpermute_idx(icp_handle *h, const uint32_t *d_slot_idx, size_t n, uint32_t *d_out);
hipError_t launch_nn_grid(icp_handle *h, const double *d_src, size_t n, const Pose *T, double *d_a,
                          double *d_b, uint32_t *d_idx);
// ... with the pose read from device memory (nn_grid.hip; *launched = false: not this time, nothing enqueued)
hipError_t launch_nn_grid_ahead(icp_handle *h, const double *d_src, size_t n, const AheadPose *d_pose, double *d_a,
                                double *d_b, uint32_t *d_idx, bool *launched);

// one inner Gauss-Newton iteration's device work; results land in h->ws.h_res after the
// stream is synchronised
hipError_t launch_weighted_gn(icp_handle *h, const double *d_a, const double *d_b, size_t n,
                              const Pose &T);
// the same through the short pipeline (7 launches, or 3 when n <= kSelCap); sets
// h_res->overflow when the caller has to redo the evaluation with launch_weighted_gn
hipError_t launch_weighted_gn_fast(icp_handle *h, const double *d_a, const double *d_b, size_t n,
                                   const Pose &T);
hipError_t launch_weighted_gn_pull(icp_handle *h, const double *d_a, const double *d_b, size_t n,
                                   const Pose &T);
// the whole Icp::estimate of a small cloud in one launch (gn_fast.hip); *status = -1: not served
hipError_t launch_tiny_estimate(icp_handle *h, const double *d_src, size_t n, const Pose &T0, size_t max_iter,
                                Pose *out, uint32_t *d_last_idx, uint32_t *inner_iters, int *status);
// three launches around a predicted window (gn_win.hip); h_res->overflow == 2 when it missed
bool window_usable(const icp_handle *h, size_t n, WinParams *P, int kind = 2, bool any_n = false,
                   double f_override = 0.);
hipError_t launch_weighted_gn_win(icp_handle *h, const double *d_a, const double *d_b, size_t n,
                                  const Pose &T, const WinParams &P);
// filed candidates (gn_win.hip, round 5): can the workgroups stage the members of these windows; two evaluations in
// two launches on one stream
bool bkt_fits(size_t n, const WinParams &P);
bool bkt_fits_rank(size_t n_total, const WinParams &P);
hipError_t launch_bkt_pair(icp_handle *h, hipStream_t s, GnCtx &first, const double *a1, const double *b1, const WinParams &P1,
                           bool ahead_on, const Pose &ahead_outer, GnCtx &second, const double *a2, const double *b2,
                           const Pose &T2, const WinParams &P2, size_t n);
// the pipelined sharded evaluation (gn_win.hip: k_win_pick_shard; pipe.hip drives it)
struct ShardPickRank {
  icp_handle *h;
  int rank, b0, nbl;          // the rank's tree blocks [b0, b0 + nbl)
  size_t n_local;
  const double *a[2], *b[2];  // its pairs of the (up to two) evaluations
};
struct ShardPickEval {
  bool alt_ctx;   // which of the handle's two evaluation contexts carries it
  Pose T;         // inner pose
  WinParams P;
  bool ahead_on;  // its finishing workgroup leaves "update x outer" for the run-ahead search
  Pose outer;
};
hipError_t launch_shard_evals(const ShardPickRank *rk, int nranks, int world, int B, size_t n_total, unsigned gen0,
                              const ShardPickEval *ev, int nevals);
// ... and the host loop around it (pipe.hip): a rank of the pipelined sharded registration
struct PipeRank {
  icp_handle *h;
  const double *d_src;  // the rank's points (a slice of the fold order; its search snapshot is prepared)
  size_t n_local;
  int rank, b0, nbl;    // its tree blocks [b0, b0 + nbl)
  uint32_t *d_idx;      // where the LAST search of the call leaves its correspondences (local order), or null
};
int pipe_run(PipeRank *rk, int nranks, int world, size_t n_total, Pose *T_io, size_t *it_io, size_t max_iter,
             uint32_t *inner_iters, int *why);
// n > 4M: the window is found in two passes (gn_win.hip, "refined windows"); the host part of the
// orchestration (two waits) lives in api.hip:wgn_step
constexpr size_t kRefineListCap = 1u << 21;  // expected: ~4e5 per dimension
constexpr size_t kRefineSample = 1u << 18;  // standard error of its median: 0.0025 sigma (1M: 0.0012, 60 us more)
double window_half_width(size_t n, bool wide);
bool refine_applies(size_t n);
bool make_window(const double med[2], const double sigma[2], double f, WinParams *P);
hipError_t launch_sample_pairs(icp_handle *h, const double *d_a, const double *d_b, size_t n);
hipError_t launch_win_first_pass(icp_handle *h, const double *d_a, const double *d_b, size_t n, const Pose &T,
                                 const WinParams &P1);  // + copy of the histograms into h->ws.h_whist
bool refine_window(const uint32_t *hist, size_t n, const WinParams &P1, WinParams *P2);
hipError_t launch_win_second_pass(icp_handle *h, const double *d_a, size_t n, const Pose &T, const WinParams &P2);
// sharded evaluation (shard.hip)
void shard_geometry(size_t n_total, int rank, int world, int *b0, int *b1, int *blocks, size_t *n_local);
hipError_t launch_shard_copy(icp_handle *h, const void *src, void *dst, size_t n_total, int rank, int world,
                             unsigned words, bool take);
hipError_t launch_shard_take_perm(icp_handle *h, const void *src, const uint32_t *perm, void *dst, size_t n_total, int rank, int world,
                                  unsigned words);
size_t shard_cand_bytes();
int shard_part_rows(int world);
size_t shard_part_bytes(int world);
size_t shard_exchange_bytes(int world);  // what a rank hands to the others: candidates + block sums
hipError_t shard_launch_hist(icp_handle *h, const double *d_a, const double *d_b, size_t n_local, const Pose &T,
                             const WinParams &P, int blocks_local);
hipError_t shard_launch_compact(icp_handle *h, size_t n_local, size_t n_total, const WinParams &P, int world,
                                int blocks_local, void *d_out);
hipError_t shard_launch_finish(icp_handle *h, const void *d_exch_all, int world, size_t n_total, int blocks_total,
                               double *d_ordered);
hipError_t shard_launch_status(icp_handle *h, int rc);
// ... the same from one pointer per rank (peer memory read in place), and the flag exchange of icp_create_multi
hipError_t shard_launch_finish_ptrs(icp_handle *h, const void *const *exch_ptrs, int world, size_t n_total,
                                    int blocks_total, double *d_ordered);
}  // namespace icp
int icp_shard_eval_finish_ptrs(icp_handle *h, const void *const *part_ptrs, double delta[3], double *huber_err);
namespace icp {
hipError_t multi_unpermute(hipStream_t s, const uint32_t *in, const uint32_t *perm, size_t n, uint32_t *out);
hipError_t multi_signal(hipStream_t s, unsigned *flag, unsigned value);
hipError_t multi_wait(hipStream_t s, const unsigned *const *flags, int world, unsigned value, unsigned *err);
hipError_t multi_sum_hist(hipStream_t s, const void *const *hists, int world, uint32_t *out);
hipError_t multi_put_pairs(hipStream_t s, const double *a_loc, const double *b_loc, size_t n_total, int rank, int world,
                           double *a_full, double *b_full);
// EXTENSION: point-to-plane residuals (p2plane.hip)
hipError_t launch_target_normals(icp_handle *h, int k, double *d_normals, size_t first = 0);
hipError_t launch_p2pl_gather(icp_handle *h, const double *d_src, size_t n, const Pose &T, const uint32_t *d_idx,
                              const double *d_normals, void *d_pairs);
hipError_t launch_p2pl_eval(icp_handle *h, const void *d_pairs, size_t n, const Pose &T, double *d_fa, double *d_fb);
size_t p2pl_pair_bytes();
// unweighted accumulation (gauss_newton_update / error / huber_error)
hipError_t launch_plain_gn(icp_handle *h, const double *d_a, const double *d_b, size_t n,
                           const Pose &T);

}  // namespace icp
