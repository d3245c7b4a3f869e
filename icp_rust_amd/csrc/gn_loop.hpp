// Interface of the one-launch inner loop (gn_loop.hip) towards the host loop in api.hip.
#pragma once
#include "common.hpp"

namespace icp {

constexpr int kLoopMaxK = 8;  // pairs a thread of the reduction tree keeps in registers

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
  int status;          // 0 ok, 1 evaluation `it` not served (window missed / no window / rotation out of sin-cos range),
                       // 3 NaN residual, 5 a grid barrier timed out (launch not resident)
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

bool gn_loop_applies(size_t n);
size_t gn_loop_partials_doubles();
hipError_t launch_gn_loop(icp_handle *h, const LoopArgs &args);

}  // namespace icp
