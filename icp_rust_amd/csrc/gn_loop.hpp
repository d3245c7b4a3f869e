// Interface of the one-launch inner loop (gn_loop.hip) towards the host loop in api.hip.
#pragma once
#include "common.hpp"

namespace icp {

constexpr int kLoopMaxK = 8;  // pairs per thread of the reduction tree that a launch keeps in LDS (2^20 pairs in all)

// Device-resident control block of the launch; all zero between launches (the last workgroup to leave resets it).
// One 128-byte line per word that is polled or hit by atomics.
struct LoopCtl {
  // one word per workgroup and barrier of an evaluation: (evaluation number | payload << 16), written once per
  // evaluation by its owner, polled by everybody (gn_loop.hip: flag_barrier)
  unsigned long long flag1[kReduceMaxBlocks];
  unsigned long long flag2[kReduceMaxBlocks];
  unsigned abort[32];            // a grid barrier timed out: every workgroup leaves
  unsigned done[32];             // workgroups that have left the launch
  unsigned nan_flag[32];
  unsigned list_cnt[2][4][32];   // [parity of the evaluation][med x, med y, ring x, ring y]
};

// What the launch hands to the host (pinned, coherent): the loop's state and the statistics the host's window
// predictions are made from.
struct LoopResult {
  Pose Ti;             // inner pose
  double prev_error;
  double med[3][2];    // exact medians of the launch's first / second / most recent evaluation
  double sigma[3][2];
  unsigned applied;    // updates applied so far (src/lib.rs:81)
  unsigned it;         // not finished: the evaluation the host resumes with (state = before that evaluation)
  unsigned evals;      // evaluations served by this launch
  unsigned rounds;     // evaluation rounds it ran: a repeated evaluation (missed window, widest windows) counts twice
  int status;          // 0 ok; 1 evaluation `it` not served (its window missed twice); 2 evaluation `it - 1` was served and
                       // applied, evaluation `it` has no usable window; 3 NaN residual; 4 evaluation `it` not served: its
                       // rotation lies outside the restated range of sin / cos (the host applies it); 5 a grid barrier
                       // gave up waiting (launch not resident, or a peer of a sharded launch raised abort)
  int finished;        // the loop ended inside the launch (break test, None, or ICP_INNER_MAX_ITER)
  unsigned seq;        // written last, system scope
};

struct LoopArgs {
  const double2 *a, *b;  // matched pairs in fold order
  unsigned n;
  unsigned it0, applied0;  // resume point: evaluation index and updates applied before it
  Pose T0;
  double prev_error0;
  WinParams PA;          // window of the launch's first evaluation
  WinParams PB;          // ... of its second, when the host has a prediction of its own for it (pb_valid)
  int pb_valid;
  double f_next;         // half-width (in sigmas) of the windows the launch centres on its own previous evaluation
  uint32_t *whist;       // 2 (parity) x 2 x kWinBins, zero at rest
  double *wmed, *wring;  // 2 x kWinCapMed, 2 x kWinCapRing
  double *partials;      // 2 (parity) x kReduceMaxBlocks x (kNSum + 1) block sums, then 2 x (kNSum + 1) folded totals
  LoopCtl *ctl;
  LoopResult *res;
  unsigned seq;
};

// ---- the same loop over the ranks of a sharded registration ---------------------------------------------------
// Rank r owns the reduction-tree blocks [b0, b1) (shard.hip) and launches one workgroup per block; what crosses the
// ranks travels through INBOXES: one per rank, in that rank's memory, mapped by every peer (peer access between the
// devices of one process, hipIpc between processes, plain pointers between virtual ranks on one device).  A producer
// writes its bytes into every rank's inbox and then a flag word; a consumer only ever polls and reads ITS OWN inbox.
// Flag words carry (generation | payload << 32); generations grow over the launches of a handle (launch number x 1024 +
// evaluation), so nothing has to be reset between launches and a rank that is one launch ahead cannot be mistaken.
constexpr int kLoopCandPerBlock = 2 * kWinBlkMed + 2 * kWinBlkRing;  // doubles a workgroup may contribute per evaluation
// ---- the PIPELINED sharded evaluation (round 6: gn_win.hip k_win_pick_shard, pipe.hip) -------------------------------
// A rank's outer iteration in the steady state of a registration (the inner loop applies one update) is the one-GPU
// pipeline -- search -> the two evaluations' first launches on the rank's own points -> ONE finishing workgroup per
// evaluation -- and that finishing workgroup is where the ranks meet: it pushes the rank's window histogram and block
// sums into every inbox, waits for every rank's, resolves the bins from the summed counts (every rank the same), picks
// ITS candidates out of its own segments, pushes them, waits for every rank's, and selects / folds / solves like one
// GPU.  One slot per evaluation in flight; kPipeBufs of them, taken round robin by the evaluation's generation: a
// rank can be at most one evaluation pair ahead of the slowest (it needs that rank's pushes to finish its own).
constexpr int kPipeBufs = 4;
constexpr int kPipeCandPerRank = 2 * kWinCapMed + 2 * kWinCapRing;  // med x | med y | ring x | ring y, full capacity each
struct PipeSlot {
  unsigned long long flag_hist[kShardMaxWorld];  // rank s: histogram + block sums of generation g are in; payload: NaN | files unusable << 1
  unsigned long long flag_cand[kShardMaxWorld];  // rank s: candidates (and their counts) of generation g are in
  unsigned cand_cnt[kShardMaxWorld][8];          // {med x, med y, ring x, ring y, this rank could not list its candidates}
  uint32_t hist[kShardMaxWorld][2 * kWinBins];   // every rank's counts, pushed
  double rows[kTreeMaxBlocks][kNSum + 1];        // block sums by global block
  double cand[kShardMaxWorld][kPipeCandPerRank];
};
struct LoopInbox {
  unsigned long long flag_block[kReduceMaxBlocks];  // own workgroups only: phase A of evaluation g is out (local barrier)
  unsigned long long flag_rank[kShardMaxWorld];     // rank s has pushed its histogram of evaluation g
  unsigned long long flag_cand[kReduceMaxBlocks];   // by global block: candidates + block sum of evaluation g are in; payload = counts
  unsigned abort[32];
  unsigned done[32];
  uint32_t hist_local[2][2 * kWinBins];                   // this rank's counts (its workgroups' atomics), by parity
  uint32_t hist_from[kShardMaxWorld][2][2 * kWinBins];    // ... and every rank's, pushed
  double partials[2][kReduceMaxBlocks][kNSum + 1];        // block sums by global block, by parity
  double cand[kReduceMaxBlocks][kLoopCandPerBlock];       // med x | med y | ring x | ring y of every workgroup
  unsigned long long probe[kShardMaxWorld];               // transport probe (icp_loop_transport_probe): rank s' last token
  PipeSlot pipe[kPipeBufs];                               // the pipelined evaluation's exchanges (above)
};

struct LoopShardArgs {
  int rank, world;
  int b0;                // (set by the kernel) first tree block of the workgroup's rank; its workgroup j is tree block b0 + j
  int blocks_total;      // reduce_geometry(n_total)
  int first_block[kShardMaxWorld + 1];  // tree blocks of rank s: [first_block[s], first_block[s + 1])
  unsigned gen_base;     // launch number x 1024 (a launch runs fewer rounds than that)
  unsigned eval_base;    // evaluations the handle's earlier launches ran (parity of the double buffers continues)
  LoopInbox *inbox[kShardMaxWorld];  // every rank's inbox as mapped here; inbox[rank] is this rank's own
};

bool gn_loop_applies(size_t n);
bool gn_loop_shard_applies(size_t n_total, int world);
// a.n = the points of the WHOLE registration; a.a / a.b = this rank's pairs (shard.hip: compact, fold order)
// each rank's pairs and pinned result block (indexed by rank)
struct LoopRankPtrs {
  const double2 *a[kShardMaxWorld], *b[kShardMaxWorld];
  LoopResult *res[kShardMaxWorld];
};
// `ranks` ranks, s.rank first, in one launch on h's stream (1 between devices and processes; all of them for ranks that
// share a device)
hipError_t launch_gn_loop_shard(icp_handle *h, const LoopArgs &a, const LoopShardArgs &s, const LoopRankPtrs &ptrs, int ranks);
size_t gn_loop_partials_doubles();
hipError_t launch_gn_loop(icp_handle *h, const LoopArgs &args);
// ping-pong over the connected inboxes: `rounds` tokens written into every peer's inbox and awaited from every peer in
// the own one (bounded wait); *d_ok (device word) = 1 if every token of every round arrived
hipError_t launch_loop_probe(icp_handle *h, int rank, int world, void *const *inboxes, unsigned base, unsigned rounds,
                             unsigned *d_ok);

}  // namespace icp
