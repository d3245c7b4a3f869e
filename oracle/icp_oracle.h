/*
 * icp_oracle.h -- CPU restatement of tier4/icp_rust's ICP path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle: a single-threaded, plain-C, IEEE-double transcription of
 * the reference crate's algorithm in the reference's own operation order
 * (compile with -ffp-contract=off, no fast-math).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product library (libicp_mi355x.so)
 * never links, loads or calls anything in oracle/.
 *
 * Pinning status: every function is checked against the known-answer tests the reference
 * holds for it (tests/test_oracle_kat.py restates all 24 of them, literals included).
 * The reference itself (Rust, nightly, un-vendored git dependencies) cannot be built in
 * this image, so there is no oracle/_ref/.  One piece is PARITY UNPINNED: the exact-NN
 * tie rule and the d^2 summation order live in the un-vendored, un-pinned crate
 * `nearest_neighbor` (Cargo.toml:22-25: git branch "main", no rev, Cargo.lock ignored).
 * Contract adopted here: d^2 = ((dx*dx + dy*dy) + dz*dz), no FMA, ties -> lowest index.
 *
 * All citations are file:line under /root/reference.
 */
#ifndef ICP_ORACLE_H
#define ICP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Transform{rot: Rotation2, t: Vector2} (src/transform.rs:6-10).  Rotation2 wraps a
 * column-major 2x2 (nalgebra ArrayStorage), hence the field order r00, r10, r01, r11. */
typedef struct { double r00, r10, r01, r11, tx, ty; } orc_pose;

#define ORC_OK 0
#define ORC_NONE 1       /* the reference returns None                       */
#define ORC_EMPTY_DST 2  /* the reference panics: index.unwrap() lib.rs:122  */
#define ORC_NAN 3        /* the reference panics: partial_cmp().unwrap()     */

/* constants: src/lib.rs:32,60,61; src/stats.rs:42 */
#define ORC_HUBER_K 1.345
#define ORC_DELTA_NORM_THRESHOLD 1e-6
#define ORC_INNER_MAX_ITER 200
#define ORC_PPF34 1.482602218505602

/* --- so2 / se2 / Transform (src/so2.rs, src/se2.rs, src/transform.rs) ------------- */
/* f64::sin / f64::cos as the reference's no_std build evaluates them (include/icp_trig.h) */
double orc_sin(double x);
double orc_cos(double x);
void orc_so2_exp(double theta, double m_colmajor[4]);                /* so2.rs:23-31 */
double orc_so2_log(const double m_colmajor[4]);                      /* so2.rs:19-21 */
void orc_se2_calc_rt(const double param[3], orc_pose *out);          /* se2.rs:21-41 */
void orc_se2_exp(const double param[3], double m3_rowmajor[9]);      /* se2.rs:43-52 */
void orc_se2_log(const double m3_rowmajor[9], double param[3]);      /* se2.rs:54-77 */
void orc_se2_get_rt(const double m3_rowmajor[9], double rot_rowmajor[4], double t[2]); /* se2.rs:11-19 */
void orc_transform_new(const double param[3], orc_pose *out);        /* transform.rs:13-16 */
void orc_transform_identity(orc_pose *out);                          /* transform.rs:34-39 */
void orc_transform_apply(const orc_pose *T, const double p[2], double out[2]); /* transform.rs:22-24 */
void orc_transform_inverse(const orc_pose *T, orc_pose *out);        /* transform.rs:26-32 */
void orc_transform_mul(const orc_pose *lhs, const orc_pose *rhs, orc_pose *out); /* transform.rs:42-51 */
void orc_transform_xy(const orc_pose *T, const double p[3], double out[3]);     /* lib.rs:52-57 */

/* --- norm / huber / linalg / stats ---------------------------------------------- */
double orc_norm_squared(const double *m_colmajor, size_t nrows, size_t ncols); /* norm.rs:8-17 */
double orc_norm(const double *m_colmajor, size_t nrows, size_t ncols);         /* norm.rs:19-21 */
double orc_huber_rho(double e, double k);                            /* huber.rs:6-15  */
double orc_huber_drho(double e, double k);                           /* huber.rs:17-26 */
int orc_inverse3x3(const double m_rowmajor[9], double out_rowmajor[9]); /* linalg.rs:3-29 */
int orc_median(double *v, size_t n, double *out);                    /* stats.rs:11-28 (reorders v) */
int orc_mad(double *v, size_t n, double *out);                       /* stats.rs:30-37 */
int orc_standard_deviation(double *v, size_t n, double *out);        /* stats.rs:39-47 */
int orc_calc_stddevs(const double *r, size_t n, size_t dim, double *out); /* stats.rs:49-60; r is n x dim AoS */

/* --- estimator (src/lib.rs:34-50, 59-84, 176-261); a,b are n x 2 AoS ---------- */
void orc_residual(const orc_pose *T, const double s[2], const double d[2], double out[2]); /* lib.rs:34-36 */
double orc_error(const orc_pose *T, const double *a, const double *b, size_t n);           /* lib.rs:38-43 */
double orc_huber_error(const orc_pose *T, const double *a, const double *b, size_t n);     /* lib.rs:45-50 */
int orc_gauss_newton_update(const orc_pose *T, const double *a, const double *b, size_t n,
                            double delta[3]);                                              /* lib.rs:191-216 */
int orc_weighted_gauss_newton_update(const orc_pose *T, const double *a, const double *b,
                                     size_t n, double delta[3]);                           /* lib.rs:218-261 */
/* returns the number of times `transform` was updated (inner iterations applied) */
int orc_estimate_transform(const double *a, const double *b, size_t n, orc_pose *out);     /* lib.rs:59-84 */

/* summation-order variant used ONLY to prove the device reduction tree: identical to
 * the two functions above except that the N-term sums (jtj, jtr, huber error) are added
 * in the fixed tree order documented in DESIGN.md "GN reduction order" instead of the
 * reference's left fold.  reduce_blocks/reduce_threads select the tree. */
int orc_weighted_gauss_newton_update_tree(const orc_pose *T, const double *a, const double *b,
                                          size_t n, int reduce_blocks, int reduce_threads,
                                          double delta[3], double *huber_err);

/* --- exact nearest neighbour (replaces nearest_neighbor::KdTree, lib.rs:26,99,121,141,164) */
/* brute force, O(n*m); dim = 2|3; q,dst AoS */
int orc_nn_brute(const double *dst, size_t m, int dim, const double *q, size_t n, uint32_t *idx);
typedef struct orc_kdtree orc_kdtree;
orc_kdtree *orc_kdtree_build(const double *dst, size_t m, int dim);  /* KdTree::new(dst, 1) */
void orc_kdtree_free(orc_kdtree *t);
void orc_set_threads(int threads); /* queries of one kd search over host cores (default 1, as the reference) */
int orc_kdtree_search(const orc_kdtree *t, const double *q, size_t n, uint32_t *idx);

/* --- ICP driver (src/lib.rs:91-174) ------------------------------------------------ */
/* use_kdtree: 0 = brute force NN, 1 = kd-tree NN (same result, ties -> lowest index).
 * sum_mode: 0 = reference left fold; 1 = device reduction tree (see above).
 * last_idx (nullable, n): correspondences of the last outer iteration.
 * inner_iters (nullable, max_iter): inner updates applied per outer iteration.   */
typedef struct {
  int use_kdtree;
  int sum_mode;
  int reduce_blocks;
  int reduce_threads;
} orc_icp_opts;
int orc_icp_estimate(int dim, const double *dst, size_t m, const double *src, size_t n,
                     const orc_pose *init, size_t max_iter, const orc_icp_opts *opts,
                     orc_pose *out, uint32_t *last_idx, uint32_t *inner_iters);
/* same, with a pre-built tree (Icp{2,3}d::new once per frame, estimate many times) */
int orc_icp_estimate_tree(const orc_kdtree *t, const double *dst, size_t m, const double *src,
                          size_t n, const orc_pose *init, size_t max_iter,
                          const orc_icp_opts *opts, orc_pose *out, uint32_t *last_idx,
                          uint32_t *inner_iters);

/* halves of orc_weighted_gauss_newton_update_tree for checking a sharded evaluation (see icp_oracle.c) */
int orc_wgn_tree_partials(const orc_pose *T, const double *a, const double *b, size_t n_local, int blocks_local,
                          int threads, double *out_blocks_x_19);
int orc_wgn_tree_fold(const double *partials_blocks_x_19, int blocks, int threads, const double stddevs[2],
                      double delta[3], double *huber_err);

/* EXTENSION CHECKER (no reference counterpart, no parity claim): point-to-plane residuals as
 * include/icp_mi355x.h section 7 defines them; see the block comment in icp_oracle.c */
int orc_p2pl_normals(const double *dst, size_t m, int k, double *normals_out);
int orc_p2pl_normals_range(const double *dst, size_t m, size_t first, int k, double *normals_out);
int orc_p2pl_estimate(const orc_kdtree *tree, const double *dst, size_t m, const double *normals,
                      const double *src, size_t n, const orc_pose *init, size_t max_iter, orc_pose *out,
                      uint32_t *last_idx, uint32_t *inner_iters);

#ifdef __cplusplus
}
#endif
#endif
