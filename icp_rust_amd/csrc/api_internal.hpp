// What the translation units of the C ABI share (api.hip: handles, stage calls, the loops of Icp::estimate; api_shard.hip:
// the sharded registration; api_ext.hip: the extensions beyond the reference).  Internal: nothing here is exported
// (csrc/exports.map).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "common.hpp"
#include "gn_loop.hpp"

#define HIP_TRY(expr)                                      \
  do {                                                     \
    hipError_t e__ = (expr);                               \
    if (e__ != hipSuccess) return ::icp::api::map_hip(e__); \
  } while (0)
#define ICP_TRY_RC(expr)             \
  do {                               \
    const int rc__ = (expr);         \
    if (rc__ != ICP_OK) return rc__; \
  } while (0)

namespace icp {
namespace api {

int map_hip(hipError_t e);
// the host's wait for a result block (polls its sequence number, then the stream the work was enqueued on)
hipError_t wait_seq(icp_handle *h, volatile unsigned *seq, unsigned want, hipStream_t stream = nullptr);
hipError_t wait_result(icp_handle *h, hipStream_t stream = nullptr);
// weighted_gauss_newton_update + huber_error on device pairs through whatever pipeline serves (api.hip: wgn_step)
int wgn_step(icp_handle *h, const double *d_a, const double *d_b, size_t n, const Pose &T, double delta[3], double *huber_err,
             bool pre_launched = false, int kind = 2);
int resolved_nn_mode(const icp_handle *h);
// check_input_size, src/lib.rs:186-189
inline bool input_size_ok(size_t n) { return n > 0 && n >= 2; }
// what the next evaluation's window is centred on: this evaluation's exact median and sigma (api.hip)
void record_statistics(Workspace &w, int kind, bool has_median, const GnResult &r);

// ---- the one-launch inner loop (gn_loop.hip), host side ----
// What a launch of the device-resident loop is planned with, and what its result is read against: the window
// predictions for its first two evaluations (taken from the handle's per-kind history, common.hpp: Workspace::win_kind).
struct LoopPlan {
  LoopArgs A;
  int first_kind = 0, second_kind = 1, it0 = 0;
  bool own0 = false;
  double p_med[2][2], p_sigma[2][2];
  int kind_of(int i) const { return i == 0 ? first_kind : (i == 1 ? second_kind : 2); }
};
bool loop_plan(icp_handle *h, size_t n, int it, int first_kind, int second_kind, bool hints, LoopPlan *pl);
int loop_finish(icp_handle *h, const LoopPlan &pl, const LoopResult *res, Pose *T, double *prev_error, uint32_t *applied,
                int *it, bool *finished);
hipError_t ensure_loop(icp_handle *h);
void loop_timed_out(Workspace &w);
void free_loop_inbox(Workspace &w);  // (api_shard.hip)
void free_loop_plan(void *p);

}  // namespace api
hipError_t launch_sel_init(icp_handle *h, size_t n);  // (gn.hip)
hipError_t launch_stddevs(icp_handle *h, const double *d_a, const double *d_b, size_t n, const Pose &T);
}  // namespace icp
