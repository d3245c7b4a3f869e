/* Observability of libicp_mi355x.so: which pipeline served what, timing hooks of the benchmark.  Test and profiling
 * infrastructure -- NOT part of the drop-in boundary (include/icp_mi355x.h, sections 1-7, is what a binding of the
 * reference's API needs); the symbols are exported by the product library so that the -m gpu tests can prove which
 * path ran.  No result depends on any of them. */
#ifndef ICP_MI355X_DEBUG_H
#define ICP_MI355X_DEBUG_H

#include "icp_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Observability for tests: which pipeline served the weighted Gauss-Newton evaluations of
 * this handle (NULL: the scratch handle behind the free functions) since it was created.
 * out[0] evaluations started with the three-launch window pipeline, out[1] how many of those
 * missed their window and were repeated, out[2] evaluations by the seven-launch (or
 * single-workgroup) pipeline, out[3] by the general radix-select path; out[4] / out[5]
 * speculative searches of icp_estimate[_device] whose pose was confirmed / discarded.  Every
 * path returns the same bits; the counters only show that a test exercised what it meant to. */
int icp_gn_path_counters(icp_handle *h, uint64_t out[6]);

/* ... and the one-launch inner loop (gn_loop.hip: the whole estimate_transform loop, src/lib.rs:59-84, of a pair set of
 * up to 2^20 in one launch): out[0] launches, out[1] evaluations they served (counted in out[0] of
 * icp_gn_path_counters as well), out[2] launches that handed an evaluation back to the host-stepped pipelines. */
int icp_gn_loop_counters(icp_handle *h, uint64_t out[3]);
/* launches of the one-launch inner loop that were not fully resident and gave up their bounded wait (2 ms without a
 * workgroup arriving at the grid barrier; 3 s between the ranks of a sharded launch).  After one, the handle steps its
 * next 64 inner loops from the host and then tries the launch again. */
int icp_gn_loop_timeouts(icp_handle *h, uint64_t *out);
/* outer iterations of icp_estimate[_device] that were not run because the one before them had left the pose as it found
 * it, bit for bit: every later iteration would repeat it (src/lib.rs:105-130 is a function of the pose and the clouds),
 * so only the last one -- which reports the correspondences -- still runs.  The inner counts of the skipped ones are 0. */
int icp_fixed_point_skips(icp_handle *h, uint64_t *out);

/* ... and the run-ahead searches of icp_estimate[_device] (a search enqueued behind the pre-launched first evaluation
 * of the next iteration, its pose derived on the device): out[0] the host derived the same pose bit for bit and took
 * the pairs, out[1] it derived another (the search was repeated for the host's pose).  A run-ahead search behind an
 * evaluation that missed its window finds no pose on the device, does not run and is counted in neither. */
int icp_run_ahead_counters(icp_handle *h, uint64_t out[2]);

/* Observability: certified matches (the searches of an estimate call after the first, beyond 65 536 source points:
 * a query whose previous match is provably still its nearest neighbour -- it has moved less than the margin the
 * last walk left it -- is not searched again).  out[0] = searches that checked certificates
 * since the handle was created, out[1] = queries whose certificate failed in the last of them (searched as ever).
 * (ICP_NN_NO_CERT=1 searches every query every time -- in the development build libicp_mi355x_exp.so only: the product
 * library reads no tuning variables) */
int icp_nn_cert_counters(icp_handle *h, uint64_t out[2]);

int icp_single_launch_counters(icp_handle *h, uint64_t out[3]);

/* Observability: out[0] = appends served incrementally (the sorted records of the search grid move up by their cells'
 * shifts and the new ones fill the gaps: possible while every new point lies within half a cell of the grid's box and
 * the cloud has grown by less than half since the grid's cell size was chosen), out[1] = appends that rebuilt the grid.
 * Either way the handle afterwards answers like a fresh handle on the concatenated cloud (in the sense stated at the
 * top of this section). */
int icp_grid_append_counters(const icp_handle *h, uint64_t out[2]);

int icp_multi_counters(const icp_multi *M, uint64_t out[2]);

/* ... and of the one-launch inner loop across the ranks (section 5b): out[0] launches (per rank), out[1] evaluations they
 * served, out[2] launches that handed an evaluation back to the stage calls.  ICP_NO_GN_LOOP=1: stage calls only; an icp_multi whose launches once gave up
 * waiting for each other (ranks that share a device need all their workgroups running at once) serves through the stage
 * calls from then on, silently: the results are the same bits either way. */
int icp_multi_loop_counters(const icp_multi *M, uint64_t out[3]);

/* ... and of the pipelined sharded registration (section 5c): outer iterations it served over the life of `M`; per
 * handle: out[0] iterations served, out[1] hand-backs, out[2] waits for a peer that ran out, out[3] run-ahead searches
 * whose pose the host confirmed. */
int icp_multi_pipe_iterations(const icp_multi *M, uint64_t *out);
int icp_pipe_counters(icp_handle *h, uint64_t out[4]);

/* enable = 0: icp_estimate[_device] on this handle runs every one of its max_iter outer iterations, also those behind a
 * fixed point of the loop (the default leaves them out: they repeat the fixed point -- same pose, indices and inner
 * counts either way).  What bench.py's `converging_pair.all_twenty_run` times.  Large clouds only (the one-launch
 * registration of clouds of up to 1 024 points has its own exit). */
int icp_set_fixed_point_exit(icp_handle *h, int enable);

/* The per-call sort of the source points by target-grid cell (csrc/qsort.hip: hand-written LSD radix sort, digits of up to
 * eleven bits) alone, on host arrays: keys_out = the keys ascending, perm_out = their original indices, equal keys in
 * ascending index -- the order icp_last_fold_order documents.  bits: the keys' significant bits (1 .. 32). */
int icp_debug_sort_cells(const uint32_t *keys, size_t n, unsigned bits, uint32_t *keys_out, uint32_t *perm_out);

/* Live kernel timing for the benchmark: with enable = k > 0, HIP events bracket every
 * k-th launch of the nearest-neighbour search kernel on the handle's stream (an event pair
 * costs a few us of stream time, so the benchmark samples instead of timing every launch);
 * 0 switches it off.  icp_profile_read synchronises the stream, returns the summed device
 * time (ms) of the timed launches and their count since the last read, and clears them. */
int icp_profile_enable(icp_handle *h, int enable);
int icp_profile_read(icp_handle *h, double *nn_kernel_ms, uint64_t *nn_kernel_launches);

#ifdef __cplusplus
}
#endif

#endif /* ICP_MI355X_DEBUG_H */
