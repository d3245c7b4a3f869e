/* A consumer of the C ABI written in plain C, the way a Rust `extern "C"` binding
 * would use it: no Python, no torch, nothing but include/icp_mi355x.h and the shared library.
 * It restates the reference's end-to-end test `test_icp_3dscan` (src/lib.rs:509-551: an "L" of 21
 * points with z in {1, 2}, truth Exp(0.01, 0.01, -0.02), start = Exp(0.05, 0.01, 0.01) * truth,
 * 20 iterations, every point within 1e-3) on top of icp_create / icp_estimate / icp_destroy and the
 * pose helpers.  Exit codes: 0 passed, 1 failed, 77 no HIP device (the library has no CPU fallback).
 * Compiling this file as C is also the check that the header is C-clean. */
#include <math.h>
#include <stdio.h>

#include "icp_mi355x.h"

int main(void) {
  double src[21][3], dst[21][3];
  int i, n = 0;
  for (i = 0; i <= 10; ++i, ++n) { src[n][0] = 0.0; src[n][1] = 0.1 * i; src[n][2] = 2.0; }
  for (i = 1; i <= 10; ++i, ++n) { src[n][0] = 0.1 * i; src[n][1] = 0.0; src[n][2] = 1.0; }
  /* the reference's literals are 0.1, 0.2, ...: the decimal literals, not products */
  {
    static const double lit[11] = {0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0};
    for (i = 0; i <= 10; ++i) src[i][1] = lit[i];
    for (i = 1; i <= 10; ++i) src[10 + i][0] = lit[i];
  }
  const double truth_param[3] = {0.01, 0.01, -0.02}, noise_param[3] = {0.05, 0.010, 0.010};
  icp_pose truth, noise, init, pred;
  icp_transform_new(truth_param, &truth);
  icp_transform_new(noise_param, &noise);
  icp_transform_mul(&noise, &truth, &init);
  for (i = 0; i < 21; ++i) { /* transform_xy, src/lib.rs:52-57 */
    icp_transform_apply(&truth, src[i], dst[i]);
    dst[i][2] = src[i][2];
  }
  if (icp_abi_version() < 1) { fprintf(stderr, "unexpected ABI version %d\n", icp_abi_version()); return 1; }
  icp_handle *h = NULL;
  int rc = icp_create(&h, 3, &dst[0][0], 21, -1);
  if (rc == ICP_NO_DEVICE) { printf("no HIP device: %s\n", icp_status_string(rc)); return 77; }
  if (rc != ICP_OK) { fprintf(stderr, "icp_create: %s\n", icp_status_string(rc)); return 1; }
  uint32_t idx[21], inner[20];
  rc = icp_estimate(h, &src[0][0], 21, &init, 20, &pred, idx, inner);
  if (rc != ICP_OK) { fprintf(stderr, "icp_estimate: %s\n", icp_status_string(rc)); icp_destroy(h); return 1; }
  int bad = 0;
  for (i = 0; i < 21; ++i) {
    double p[2];
    icp_transform_apply(&pred, src[i], p);
    const double dx = p[0] - dst[i][0], dy = p[1] - dst[i][1];
    const double e = sqrt(dx * dx + dy * dy); /* z is carried through: its difference is 0 */
    if (!(e < 1e-3)) { fprintf(stderr, "point %d off by %g\n", i, e); bad = 1; }
    if (idx[i] != (uint32_t)i) { fprintf(stderr, "point %d matched to %u\n", i, idx[i]); bad = 1; }
  }
  /* an empty target cloud is the reference's panic (src/lib.rs:165), reported as a status */
  icp_handle *empty = NULL;
  if (icp_create(&empty, 3, NULL, 0, -1) != ICP_OK || icp_estimate(empty, &src[0][0], 21, &init, 1, &pred, NULL, NULL) != ICP_EMPTY_DST) {
    fprintf(stderr, "empty dst not reported\n");
    bad = 1;
  }
  icp_destroy(empty);
  /* the growing-map extension (header section 6): appending the other half equals creating the whole */
  icp_handle *half = NULL;
  uint32_t idx2[21];
  icp_pose pred2;
  if (icp_create(&half, 3, &dst[0][0], 11, -1) != ICP_OK || icp_append_targets(half, &dst[11][0], 10, NULL) != ICP_OK ||
      icp_target_count(half) != 21 || icp_estimate(half, &src[0][0], 21, &init, 20, &pred2, idx2, NULL) != ICP_OK) {
    fprintf(stderr, "append path failed\n");
    bad = 1;
  } else {
    for (i = 0; i < 21; ++i) bad |= idx2[i] != idx[i];
    bad |= pred2.r00 != pred.r00 || pred2.r10 != pred.r10 || pred2.tx != pred.tx || pred2.ty != pred.ty;
    if (bad) fprintf(stderr, "grown handle differs from the fresh one\n");
  }
  icp_destroy(half);
  icp_destroy(h);
  icp_trim_pool();
  printf(bad ? "FAILED\n" : "ok: test_icp_3dscan through the C ABI, pose (%.6f %.6f | %.6f %.6f), inner[0] = %u\n", pred.r00, pred.r10,
         pred.tx, pred.ty, inner[0]);
  return bad;
}
