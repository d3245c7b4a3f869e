// Exact nearest neighbour through a uniform grid built on the device.
//
// Same contract as the brute-force sweep (nn_brute.hip) and therefore the same results,
// bit for bit: d^2 = ((dx*dx + dy*dy) + dz*dz) in f64 without FMA contraction, ties ->
// lowest target index.  It replaces the reference's kd-tree (nearest_neighbor::KdTree,
// src/lib.rs:99,121,141,164) the MI355X way: instead of a pointer-chasing tree, targets
// are counting-sorted into cells (16-B screening records: f32 offsets + original index; the exact
// f64 coordinates stay in dst), a query walks the cell box of the ball around its previous match
// (warm) or Chebyshev shells of cells around its own cell (cold) and stops as soon as no
// unvisited cell can hold a closer-or-equal point.  Exactness does not depend on the cell size or on
// floating-point rounding of the cell assignment: every pruning bound is relaxed by a
// margin that is orders of magnitude above the rounding of the cell arithmetic, so a
// bound can only cause extra visits, never a missed candidate; candidates are compared
// by (d^2, original index), so the visiting order cannot change the winner.
#include "common.hpp"
#include <type_traits>

namespace icp {

__device__ __forceinline__ int cell_coord(double v, double lo, double inv_h, int n) {
  double t = floor((v - lo) * inv_h);
  t = fmin(fmax(t, 0.), (double)(n - 1));  // NaN -> 0
  return (int)t;
}

// ---------------------------------------------------------------- build ----------
__global__ void k_grid_bbox(const double *__restrict__ dst, unsigned m, int dim, double *__restrict__ part) {
  __shared__ double smn[3][256], smx[3][256];
  double mn[3], mx[3];
  for (int d = 0; d < 3; ++d) {
    mn[d] = __builtin_huge_val();
    mx[d] = -__builtin_huge_val();
  }
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256)
    for (int d = 0; d < dim; ++d) {
      const double v = dst[(size_t)i * dim + d];
      mn[d] = fmin(mn[d], v);
      mx[d] = fmax(mx[d], v);
    }
  for (int d = 0; d < 3; ++d) {
    smn[d][threadIdx.x] = mn[d];
    smx[d][threadIdx.x] = mx[d];
  }
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int d = 0; d < 3; ++d) {
        smn[d][threadIdx.x] = fmin(smn[d][threadIdx.x], smn[d][threadIdx.x + s]);
        smx[d][threadIdx.x] = fmax(smx[d][threadIdx.x], smx[d][threadIdx.x + s]);
      }
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int d = 0; d < 3; ++d) {
      part[blockIdx.x * 6 + d] = smn[d][0];
      part[blockIdx.x * 6 + 3 + d] = smx[d][0];
    }
}

__global__ void k_grid_count(const double *__restrict__ dst, unsigned m, int dim, GridParams g,
                             uint32_t *__restrict__ cell_of, uint32_t *__restrict__ cnt) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  int c[3] = {0, 0, 0};
  for (int d = 0; d < dim; ++d) c[d] = cell_coord(dst[(size_t)i * dim + d], g.lo[d], g.inv_h[d], g.n[d]);
  const uint32_t cell = ((uint32_t)c[2] * g.n[1] + c[1]) * g.n[0] + c[0];
  cell_of[i] = cell;
  atomicAdd(&cnt[cell], 1u);
}

// exclusive scan, three phases; 2048 items per block
constexpr int kScanItems = 2048;
__global__ __launch_bounds__(256) void k_scan_local(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                    unsigned n, uint32_t *__restrict__ block_tot) {
  __shared__ uint32_t wsum[4];
  const unsigned base = blockIdx.x * kScanItems + threadIdx.x * 8;
  uint32_t v[8], tot = 0;
  for (int j = 0; j < 8; ++j) {
    v[j] = (base + j < n) ? in[base + j] : 0u;
    tot += v[j];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = tot;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t wb = 0;
  for (int w = 0; w < wave; ++w) wb += wsum[w];
  uint32_t run = wb + inc - tot;
  for (int j = 0; j < 8; ++j) {
    if (base + j < n) out[base + j] = run;
    run += v[j];
  }
  if (threadIdx.x == 255) block_tot[blockIdx.x] = run;
}

__global__ __launch_bounds__(1024) void k_scan_totals(uint32_t *__restrict__ block_tot, unsigned nb,
                                                      uint32_t *__restrict__ grand_total) {
  // one block; nb <= 1024 * 16
  __shared__ uint32_t wsum[16];
  const int per = (nb + 1023) / 1024;
  uint32_t v[16], tot = 0;
  for (int j = 0; j < per; ++j) {
    const unsigned k = threadIdx.x * per + j;
    v[j] = k < nb ? block_tot[k] : 0u;
    tot += v[j];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = tot;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t wb = 0;
  for (int w = 0; w < wave; ++w) wb += wsum[w];
  uint32_t run = wb + inc - tot;
  for (int j = 0; j < per; ++j) {
    const unsigned k = threadIdx.x * per + j;
    if (k < nb) block_tot[k] = run;
    run += v[j];
  }
  if (threadIdx.x == 1023) *grand_total = run;
}

__global__ void k_scan_add(uint32_t *__restrict__ out, unsigned n, const uint32_t *__restrict__ block_tot) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += block_tot[i / kScanItems];
}

__global__ void k_grid_scatter(const double *__restrict__ dst, unsigned m, int dim,
                               const uint32_t *__restrict__ cell_of, const uint32_t *__restrict__ start,
                               uint32_t *__restrict__ cursor, GridParams g, GridPoint *__restrict__ pts) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) {
    // sentinels behind the last record: the warm searches read records in quads (four records past a run's last at most)
    if (i < m + kGridPad) pts[i] = GridPoint{__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf(), 0u};
    return;
  }
  const uint32_t cell = cell_of[i];
  const uint32_t pos = start[cell] + atomicAdd(&cursor[cell], 1u);
  GridPoint p;
  p.x = (float)(dst[(size_t)i * dim + 0] - g.lo[0]);
  p.y = (float)(dst[(size_t)i * dim + 1] - g.lo[1]);
  p.z = dim == 3 ? (float)(dst[(size_t)i * dim + 2] - g.lo[2]) : 0.f;
  p.idx = i;
  pts[pos] = p;
}

hipError_t build_grid(icp_handle *h) {
  Grid &G = h->grid;
  G.built = false;
  const unsigned m = (unsigned)h->m;
  if (m == 0) return hipSuccess;
  hipError_t e;
  hipStream_t s = h->stream;
  // 1. bounding box
  const int bb = 256;
  if (!G.t_part && (e = hipMalloc(&G.t_part, bb * 6 * sizeof(double))) != hipSuccess) return e;
  hipLaunchKernelGGL(k_grid_bbox, dim3(bb), dim3(256), 0, s, h->d_dst, m, h->dim, G.t_part);
  double part[bb * 6];
  e = hipMemcpyAsync(part, G.t_part, sizeof(part), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) return e;
  GridParams g;
  for (int d = 0; d < 3; ++d) {
    g.lo[d] = __builtin_huge_val();
    g.hi[d] = -__builtin_huge_val();
  }
  for (int b = 0; b < bb; ++b)
    for (int d = 0; d < 3; ++d) {
      g.lo[d] = fmin(g.lo[d], part[b * 6 + d]);
      g.hi[d] = fmax(g.hi[d], part[b * 6 + 3 + d]);
    }
  double ext[3], emax = 0., scale = 0.;
  for (int d = 0; d < 3; ++d) {
    if (d >= h->dim) g.lo[d] = g.hi[d] = 0.;
    if (!std::isfinite(g.lo[d]) || !std::isfinite(g.hi[d])) return hipSuccess;  // no grid: brute force serves
    ext[d] = g.hi[d] - g.lo[d];
    emax = fmax(emax, ext[d]);
    scale = fmax(scale, fmax(fabs(g.lo[d]), fabs(g.hi[d])));
  }
  if (!std::isfinite(emax)) return hipSuccess;
  // 2. cell size: ~2 targets per cell over the non-degenerate extents (measured sweet spot 2-4), <= 2^24 cells
  int k = 0;
  double vol = 1.;
  for (int d = 0; d < h->dim; ++d)
    if (ext[d] > 1e-9 * emax) {
      vol *= ext[d];
      ++k;
    }
  // ICP_GRID_OCC: targets per cell the cell size aims at (tuning knob; any value is exact)
  static const double occ = exp_env("ICP_GRID_OCC") ? atof(exp_env("ICP_GRID_OCC")) : 2.;
  double hh = k > 0 ? pow(occ * vol / (double)m, 1. / k) : 1.;
  if (!(hh > 0.) || !std::isfinite(hh)) hh = 1.;
  // ICP_GRID_FX: cells are that many times finer along x.  A row of cells along x is one
  // contiguous run of records, so finer x cells clip the runs tighter around [qx - r, qx + r]
  // without adding rows (dense surfaces put ~10 targets into a cubic cell of the average
  // occupancy; the search radius there is a fraction of the cell).
  static const double fx = exp_env("ICP_GRID_FX") ? fmax(1., atof(exp_env("ICP_GRID_FX"))) : 4.;
  // The cell size grows until the cells fit BOTH the 2^24 total and the per-axis limits (16384 along
  // x, 4096 along y / z): an elongated cloud (a corridor map: 20000 x 50 x 5 m) must not collapse
  // everything beyond the capped axis into its last cell -- results would stay exact (edge cells are
  // unbounded) but queries there would scan tens of thousands of records.
  // Only the offending axis gets coarser cells (every pruning bound uses the per-axis cell size).
  for (;;) {
    double cells = 1.;
    for (int d = 0; d < 3; ++d) {
      g.h[d] = d == 0 ? hh / fx : hh;
      const bool flat = !(d < h->dim && ext[d] > 1e-9 * emax);
      const double cap = d == 0 ? 16384. : 4096.;
      double nd = flat ? 1. : floor(ext[d] / g.h[d]) + 1.;
      if (nd > cap) {
        g.h[d] = ext[d] / (cap - 1.5);
        nd = fmin(floor(ext[d] / g.h[d]) + 1., cap);
      }
      g.n[d] = (int)nd;
      cells *= g.n[d];
    }
    if (cells <= 16777216.) break;
    hh *= 1.26;
  }
  for (int d = 0; d < 3; ++d) g.inv_h[d] = 1. / g.h[d];
  g.fx = (int)fmin(fmax(floor(fx + 0.5), 1.), 64.);
  g.scale = scale + hh;
  g.ext = (float)(emax + hh);
  // k_nn_grid_warm evaluates every pruning bound in f32, relative to the grid origin, with explicit
  // margins; it needs cell sizes and extents that f32 represents as normal numbers with headroom
  g.f32_ok = 1;
  for (int d = 0; d < 3; ++d)
    if (!(g.h[d] > 1e-10 && g.h[d] < 1e10)) g.f32_ok = 0;
  if (!(emax + hh < 1e10)) g.f32_ok = 0;
  for (int d = 0; d < 3; ++d) {
    g.hf[d] = (float)g.h[d];
    g.ihf[d] = (float)g.inv_h[d];
    g.nm1f[d] = (float)(g.n[d] - 1);
  }
  G.p = g;
  G.ncell = (uint32_t)g.n[0] * g.n[1] * g.n[2];
  // 3. counting sort of the targets by cell
  const unsigned nscan = G.ncell + 1;
  const unsigned nb = (nscan + kScanItems - 1) / kScanItems;
  do {
    if ((e = reserve(G.t_cell_of, G.cap_tcell, (size_t)m)) != hipSuccess) break;
    if ((e = reserve(G.t_cnt, G.cap_tcnt, (size_t)nscan)) != hipSuccess) break;
    if ((e = reserve(G.t_btot, G.cap_tbtot, (size_t)nb + 1)) != hipSuccess) break;
    if ((e = reserve(G.d_start, G.cap_start, (size_t)nscan)) != hipSuccess) break;
    if ((e = reserve(G.d_pts, G.cap_pts, (size_t)m + kGridPad)) != hipSuccess) break;
    uint32_t *cell_of = G.t_cell_of, *cnt = G.t_cnt, *btot = G.t_btot;
    if ((e = hipMemsetAsync(cnt, 0, (size_t)nscan * 4, s)) != hipSuccess) break;
    hipLaunchKernelGGL(k_grid_count, dim3((m + 255) / 256), dim3(256), 0, s, h->d_dst, m, h->dim, g, cell_of, cnt);
    hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(256), 0, s, cnt, G.d_start, nscan, btot);
    hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, s, btot, nb, btot + nb);
    hipLaunchKernelGGL(k_scan_add, dim3((nscan + 255) / 256), dim3(256), 0, s, G.d_start, nscan, btot);
    if ((e = hipMemsetAsync(cnt, 0, (size_t)nscan * 4, s)) != hipSuccess) break;
    hipLaunchKernelGGL(k_grid_scatter, dim3((m + kGridPad + 255) / 256), dim3(256), 0, s, h->d_dst, m, h->dim, cell_of,
                       G.d_start, cnt, g, G.d_pts);
    e = hipGetLastError();  // stream order is all the later kernels need; create_common synchronises once at the end
  } while (0);
  if (e != hipSuccess) return e;
  G.built = true;
  G.m_full = m;
  G.rcell_valid = false;  // (derived from the cell offsets by the first append that needs it)
  return hipSuccess;
}

// ---------------------------------------------------------------- append ---------
// A map gains one scan per frame (icp_append_targets).  Rebuilding the grid over ALL targets costs a pass over the
// whole cloud for the bounding box, one for the cell counts (an atomic per target) and a scatter with an atomic per
// target again: 1.2 ms per 28.8k-point append at 10M targets, as much as the registration it serves.  But the old
// records are already sorted: with the grid's box and cell size unchanged, a record of cell c simply moves up by
// shift[c] = the number of NEW points in the cells before c -- a streaming copy, no atomics -- and the new records
// fill the gap at the end of their cells.  Applicable while every new point lies inside the grid's box (the pruning
// margins of the searches assume it) and the cloud has not outgrown the cell size chosen at the last full build
// (ICP_GRID_REBUILD_GROWTH, default 1.5 x); otherwise the caller rebuilds.  Any grid gives the exact result.
// "inside": within half a cell of the box.  The pruning margins of the searches are derived for grid-relative
// coordinates of magnitude <= g.ext = the largest extent + ONE cell (build_grid), and cells at the rim of the grid
// are unbounded outwards (targets are clamped into them), so a target up to half a cell outside the box is served
// exactly like one inside it -- a registered scan of the same scene scatters around the map's faces by its noise.
__global__ void k_points_inside(const double *__restrict__ p, unsigned k, int dim, GridParams g, double tol,
                                uint32_t *flag) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  bool in = true;
  for (int d = 0; d < dim; ++d) {
    const double v = p[(size_t)i * dim + d];
    in = in && v >= g.lo[d] - tol && v <= g.hi[d] + tol;  // (NaN: false)
  }
  if (!in) atomicOr(flag, 1u);
}
// the cell of the record at every sorted position, from the cell offsets (one thread per cell)
__global__ void k_grid_fill_rcell(const uint32_t *__restrict__ start, unsigned ncell, uint32_t *__restrict__ rcell) {
  const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const uint32_t s = start[c], e = start[c + 1];
  for (uint32_t p = s; p < e; ++p) rcell[p] = c;
}
__global__ void k_grid_move(const GridPoint *__restrict__ pts, const uint32_t *__restrict__ rcell, unsigned m,
                            const uint32_t *__restrict__ shift, GridPoint *__restrict__ pts2,
                            uint32_t *__restrict__ rcell2) {
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m) return;
  const uint32_t c = rcell[p];
  const uint32_t q = p + shift[c];
  pts2[q] = pts[p];
  rcell2[q] = c;
}
__global__ void k_grid_shift_starts(const uint32_t *__restrict__ start, const uint32_t *__restrict__ shift, unsigned nscan,
                                    uint32_t *__restrict__ start2) {
  const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nscan) start2[c] = start[c] + shift[c];
}
__global__ void k_grid_insert(const double *__restrict__ tail, unsigned k, unsigned m_old, int dim,
                              const uint32_t *__restrict__ cell_new, const uint32_t *__restrict__ start,
                              const uint32_t *__restrict__ start2, uint32_t *__restrict__ cnt, GridParams g,
                              GridPoint *__restrict__ pts2, uint32_t *__restrict__ rcell2) {
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k + kGridPad) return;
  if (j >= k) {  // sentinels behind the last record (quad-aligned reads)
    pts2[m_old + j] = GridPoint{__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf(), 0u};
    return;
  }
  const uint32_t c = cell_new[j];
  // cell c: [start2[c], start2[c + 1]) -- the old records first (moved as a block), the new ones behind them
  const uint32_t pos = start2[c] + (start[c + 1] - start[c]) + (atomicSub(&cnt[c], 1u) - 1u);
  GridPoint p;
  p.x = (float)(tail[(size_t)j * dim + 0] - g.lo[0]);
  p.y = (float)(tail[(size_t)j * dim + 1] - g.lo[1]);
  p.z = dim == 3 ? (float)(tail[(size_t)j * dim + 2] - g.lo[2]) : 0.f;
  p.idx = m_old + j;
  pts2[pos] = p;
  rcell2[pos] = c;
}

hipError_t append_grid(icp_handle *h, size_t m_old_, size_t k_, bool *done) {
  *done = false;
  Grid &G = h->grid;
  static const bool off = exp_env("ICP_GRID_NO_APPEND") != nullptr;
  static const double growth = exp_env("ICP_GRID_REBUILD_GROWTH") ? atof(exp_env("ICP_GRID_REBUILD_GROWTH")) : 1.5;
  if (off || !G.built || m_old_ == 0 || k_ == 0 || G.m_full == 0) return hipSuccess;
  if ((double)(m_old_ + k_) > growth * (double)G.m_full) return hipSuccess;  // the cell size is due for a re-tune
  const unsigned m_old = (unsigned)m_old_, k = (unsigned)k_, m_new = m_old + k;
  const GridParams g = G.p;
  hipStream_t s = h->stream;
  hipError_t e;
  const double *tail = h->d_dst + m_old_ * h->dim;
  // 1. every new point inside the box?
  if (!G.d_flag && (e = hipMalloc(&G.d_flag, sizeof(uint32_t))) != hipSuccess) return e;
  if ((e = hipMemsetAsync(G.d_flag, 0, sizeof(uint32_t), s)) != hipSuccess) return e;
  // (the smallest cell size over the axes the cloud extends along; g.ext holds one cell of the LARGEST)
  double tol = __builtin_huge_val();
  for (int d = 0; d < h->dim; ++d)
    if (g.n[d] > 1) tol = fmin(tol, 0.5 * g.h[d]);
  if (!(tol < __builtin_huge_val())) tol = 0.;
  hipLaunchKernelGGL(k_points_inside, dim3((k + 255) / 256), dim3(256), 0, s, tail, k, h->dim, g, tol, G.d_flag);
  uint32_t outside = 1;
  if ((e = hipMemcpyAsync(&outside, G.d_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
  if (outside) return hipSuccess;
  // 2. cells and counts of the new points, prefix sums = the shift of every cell
  const unsigned nscan = G.ncell + 1;
  const unsigned nb = (nscan + kScanItems - 1) / kScanItems;
  if ((e = reserve(G.t_cell_new, G.cap_cell_new, (size_t)k)) != hipSuccess) return e;
  if ((e = reserve(G.t_cnt, G.cap_tcnt, (size_t)nscan)) != hipSuccess) return e;
  if ((e = reserve(G.t_btot, G.cap_tbtot, (size_t)nb + 1)) != hipSuccess) return e;
  if ((e = reserve(G.t_shift, G.cap_shift, (size_t)nscan)) != hipSuccess) return e;
  if ((e = reserve(G.d_start2, G.cap_start2, (size_t)nscan)) != hipSuccess) return e;
  if ((e = reserve(G.d_pts2, G.cap_pts2, (size_t)m_new + kGridPad)) != hipSuccess) return e;
  if ((e = reserve(G.d_rcell2, G.cap_rcell2, (size_t)m_new)) != hipSuccess) return e;
  if (!G.rcell_valid) {
    if ((e = reserve(G.d_rcell, G.cap_rcell, (size_t)m_old)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_grid_fill_rcell, dim3((G.ncell + 255) / 256), dim3(256), 0, s, (const uint32_t *)G.d_start, G.ncell,
                       G.d_rcell);
    G.rcell_valid = true;
  }
  if ((e = hipMemsetAsync(G.t_cnt, 0, (size_t)nscan * 4, s)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_grid_count, dim3((k + 255) / 256), dim3(256), 0, s, tail, k, h->dim, g, G.t_cell_new, G.t_cnt);
  hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(256), 0, s, G.t_cnt, G.t_shift, nscan, G.t_btot);
  hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, s, G.t_btot, nb, G.t_btot + nb);
  hipLaunchKernelGGL(k_scan_add, dim3((nscan + 255) / 256), dim3(256), 0, s, G.t_shift, nscan, G.t_btot);
  // 3. old records up by their cell's shift, cell offsets likewise, new records into the gaps
  hipLaunchKernelGGL(k_grid_move, dim3((m_old + 255) / 256), dim3(256), 0, s, (const GridPoint *)G.d_pts,
                     (const uint32_t *)G.d_rcell, m_old, (const uint32_t *)G.t_shift, G.d_pts2, G.d_rcell2);
  hipLaunchKernelGGL(k_grid_shift_starts, dim3((nscan + 255) / 256), dim3(256), 0, s, (const uint32_t *)G.d_start,
                     (const uint32_t *)G.t_shift, nscan, G.d_start2);
  hipLaunchKernelGGL(k_grid_insert, dim3((k + kGridPad + 255) / 256), dim3(256), 0, s, tail, k, m_old, h->dim,
                     (const uint32_t *)G.t_cell_new, (const uint32_t *)G.d_start, (const uint32_t *)G.d_start2, G.t_cnt, g,
                     G.d_pts2, G.d_rcell2);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  std::swap(G.d_pts, G.d_pts2);
  std::swap(G.cap_pts, G.cap_pts2);
  std::swap(G.d_rcell, G.d_rcell2);
  std::swap(G.cap_rcell, G.cap_rcell2);
  std::swap(G.d_start, G.d_start2);
  std::swap(G.cap_start, G.cap_start2);
  *done = true;
  return hipSuccess;
}

// ---------------------------------------------------------------- query ----------
#ifdef ICP_NN_STATS
// Diagnostic build only (make STATS=1 -> libicp_mi355x_stats.so, never the product library):
// per-launch totals of what the search actually does.  [0] queries, [1] row-bound fetches,
// [2] record batches, [3] exact evaluations, [4] sum over waves of wave-level loop steps,
// [5] sum over waves of lifetime in shader cycles, [6] waves, [7] warm queries.
__device__ unsigned long long g_nn_stats[8];
__device__ unsigned long long g_nn_hist[2][32];  // [0]: lanes by record chunks, [1]: waves by loop steps
// the warm walk: [0] wave-level row groups, [1] wave-level record chunks, [2] ... of them with an exact test in some lane,
// [3] lane row groups, [4] lane chunks, [5] lane exact tests, [6] lifetime (cycles, summed over waves), [7] waves
__device__ unsigned long long g_nn_warm[8];
__device__ unsigned long long g_nn_warm_hist[2][32];  // waves by row groups; waves by record chunks
#define NN_STAT(i, v) (st[i] += (v))
#else
#define NN_STAT(i, v) ((void)0)
#endif

// COLD: no previous matches exist (first search of a source snapshot, or unsorted queries).
// Register budget of the warm 3-D instantiation: 120 VGPRs -- three of its waves plus two waves of an
// evaluation kernel (<= 72) share a SIMD during the speculative overlap (tests/test_registers.py).
//
// L lanes per query (1, or 4 for clouds of tens of thousands of points, where a one-lane-per-query
// launch leaves most of the chip idle and lasts exactly as long as its slowest wave): the lanes of
// a query share the rows of walk_box's cell box -- lane `sub` takes rows sub, sub + L, ... in the
// order the box enumerates them -- and exchange their winners by (d^2, index) at the end.  Every
// lane prunes with the distance of a real target of the SAME query (a valid bound, merely less
// tight than the group's), every row is some lane's, and the lexicographic minimum over the lanes
// is the minimum over all records visited: same result as L = 1.
// One wave per workgroup.  The time of a wave is set by its slowest lane and varies several-fold;
// four-wave workgroups held their registers and slots until the last of the four had finished, and
// the evaluation kernels that run beside a speculative search had to wait for whole workgroups to
// retire: 256 -> 128 -> 64 threads gave 4970 -> 5440 -> 5480 iterations/s on the 1M pair (A/B on one
// box), the search alone 103.6 -> 97.2 -> 95.6 us.
// Which wave of 64 consecutive (cell-sorted) queries a workgroup takes.  Workgroups are dealt round-robin over the 8
// XCDs, each with an L2 of its own: taken in launch order, neighbouring waves -- which walk the same rows of cells
// and read the same record lines -- land on eight different L2s, and every line is fetched from the fabric up to eight
// times.  With chunk > 0 the `chunk` waves an XCD receives out of every 8 x chunk consecutive workgroups are
// CONSECUTIVE waves (a bijection inside each complete group of 8 x chunk; the tail keeps launch order), fine enough a
// grain to keep the load balanced across the XCDs (dense walls and sparse interior alternate along the sorted order).
__device__ __forceinline__ unsigned xcd_wave(unsigned block, unsigned nblocks, unsigned chunk) {
  if (chunk == 0u) return block;
  const unsigned per = 8u * chunk, full = nblocks / per * per;
  if (block >= full) return block;
  const unsigned sc = block / per, r = block % per;
  return sc * per + (r & 7u) * chunk + (r >> 3);
}

constexpr unsigned kXcdChunk = 16;  // (fetch per warm search: 112 MB in launch order, 72 / 65 / 72 / 75 MB with chunks of 8 / 16 / 32 / 64: profiles/r04_search_xcd_chunk_traffic.txt)
constexpr int kGridThreads = 64;
template <int DIM, bool XFORM, bool COLD, int L>
__global__ __launch_bounds__(kGridThreads) void k_nn_grid(const double *__restrict__ src,
                                                 const uint32_t *__restrict__ perm, unsigned n, Pose T,
                                                 GridParams g, const uint32_t *__restrict__ start,
                                                 const GridPoint *__restrict__ pts,
                                                 const double *__restrict__ dst, uint32_t *__restrict__ idx,
                                                 double2 *__restrict__ a, double2 *__restrict__ b,
                                                 const PrevMatch *__restrict__ prev, PrevMatch *prev_out,
                                                 const AheadPose *__restrict__ ahead) {
#ifdef ICP_NN_STATS
  unsigned st[8] = {1, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
  if (XFORM && ahead) {  // a search enqueued before the host knew its pose (launch_nn_grid_ahead)
    if (!ahead->valid) return;
    T = ahead->T;
  }
  const unsigned wave0 = xcd_wave(blockIdx.x, gridDim.x, kXcdChunk);
  const unsigned k = (wave0 * kGridThreads + threadIdx.x) / L;
  const unsigned sub = (wave0 * kGridThreads + threadIdx.x) % L;  // lane within the query's group (aligned: L divides 64)
  if (k >= n) return;  // whole groups leave together
  // perm != null: src is the cell-sorted copy made by prepare_queries (neighbouring lanes
  // search neighbouring cells); results go back to the original positions
  const unsigned i = perm ? perm[k] : k;
  double q[3];
  q[0] = src[(size_t)k * DIM + 0];
  q[1] = src[(size_t)k * DIM + 1];
  q[2] = DIM == 3 ? src[(size_t)k * DIM + 2] : 0.;
  if (XFORM) {  // Transform::transform, src/transform.rs:22-24
    const double nx = (T.r00 * q[0] + T.r01 * q[1]) + T.tx;
    const double ny = (T.r10 * q[0] + T.r11 * q[1]) + T.ty;
    q[0] = nx;
    q[1] = ny;
  }
  double mg[3];  // rounding margin per axis: >> ulp(cell arithmetic), << cell size
#pragma unroll
  for (int d = 0; d < 3; ++d) mg[d] = 1e-9 * (fabs(q[d]) + g.scale);

  // Candidates are screened in f32 (one 16-B record each) and only the survivors are
  // evaluated exactly.  A record holds fl32(p - lo); with qf = fl32(q - lo) every component of
  // df = qf - pf differs from the true q - p by at most ec = 2^-23 (E + |q - lo|) (three f32
  // roundings of magnitudes <= E resp. |q - lo|), so the true distance is >= |df| - sqrt(3) ec,
  // and |df|^2 is >= s32 (1 - 1e-6) for the f32-evaluated sum s32.  A candidate may be skipped
  // iff that lower bound is > sqrt(best); equivalently s32 > thr32 with thr32 rounded up.
  // Anything that could win OR TIE is therefore evaluated exactly, in f64, with the contract's
  // formula -- the screen changes which candidates are looked at, never the result.
  float qf[3];
  double ec = 0.;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const double off = (d < DIM) ? q[d] - g.lo[d] : 0.;
    qf[d] = (float)off;
    ec = fmax(ec, fabs(off));
  }
  // extent E <= 2 * g.scale; 1.2e-7 > 2^-23; the factor sqrt(3) turns the per-component bound
  // into a bound on the norm of the error vector
  ec = (ec + 2. * g.scale) * 1.2e-7 * 1.7320508075688774;

  // The running winner is (best, bi) only -- its coordinates are fetched once at the end.
  double best = __builtin_huge_val();
  uint32_t bi = 0xffffffffu;
  float thr32 = __builtin_huge_valf();
  double bx = 0., by = 0., bz = 0.;  // the winner's exact coordinates (outputs + next warm start)
  auto eval = [&](uint32_t ti, double tx, double ty, double tz) {
    NN_STAT(3, 1);
    const double ddx = q[0] - tx;
    const double ddy = q[1] - ty;
    double dd = ddx * ddx + ddy * ddy;
    if (DIM == 3) {
      const double ddz = q[2] - tz;
      dd = dd + ddz * ddz;
    }
    if (dd < best || (dd == best && ti < bi)) {
      best = dd;
      bi = ti;
      bx = tx;
      by = ty;
      bz = tz;
      const double r = sqrt(dd) + ec;
      thr32 = (float)(r * r * 1.000004);
      thr32 = thr32 * 1.000001f + 1e-37f;  // round up past the f64->f32 conversion
    }
  };
  // exact: d^2 = ((dx*dx + dy*dy) + dz*dz), ties -> lowest index.  The current winner is not
  // re-evaluated when the scan meets it again (it always passes its own screen).
  auto consider = [&](uint32_t ti) {
    if (ti == bi) return;
    eval(ti, dst[(size_t)ti * DIM + 0], dst[(size_t)ti * DIM + 1], DIM == 3 ? dst[(size_t)ti * DIM + 2] : 0.);
  };
  // L > 1: adopt the best (d^2, index) any lane of the group holds.  Called where the group's lanes
  // have reconverged (after walk_box's loop).
  auto share_best = [&]() {
    if (L == 1) return;
    // lane ^ off inside a group of at most four lanes = a quad permutation: DPP moves, no trip through the LDS crossbar
    auto quad_xor32 = [](int v, int off) -> int {
      return off == 1 ? __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true)    // quad_perm [1, 0, 3, 2]
                      : __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);  // quad_perm [2, 3, 0, 1]
    };
    auto quad_xor64 = [&](double v, int off) -> double {
      return __hiloint2double(quad_xor32(__double2hiint(v), off), quad_xor32(__double2loint(v), off));
    };
    static_assert(L <= 4, "share_best exchanges inside quads");
#pragma unroll
    for (int off = 1; off < L; off <<= 1) {
      const double ob = quad_xor64(best, off);
      const uint32_t obi = (uint32_t)quad_xor32((int)bi, off);
      const double ox = quad_xor64(bx, off), oy = quad_xor64(by, off), oz = DIM == 3 ? quad_xor64(bz, off) : 0.;
      if (ob < best || (ob == best && obi < bi)) {
        best = ob;
        bi = obi;
        bx = ox;
        by = oy;
        bz = oz;
      }
    }
    if (best < __builtin_huge_val()) {
      const double r = sqrt(best) + ec;
      thr32 = (float)(r * r * 1.000004);
      thr32 = thr32 * 1.000001f + 1e-37f;
    }
  };
  auto screen = [&](const GridPoint &t) -> float {
    const float fx = qf[0] - t.x, fy = qf[1] - t.y;
    float s = fx * fx + fy * fy;
    if (DIM == 3) {
      const float fz = qf[2] - t.z;
      s = s + fz * fz;
    }
    return s;
  };
  // kBatch records of a contiguous run in flight per lane (the tail re-reads the last record:
  // evaluating a target twice cannot change the winner, and a repeated address costs next to
  // nothing -- divergent control flow is what is expensive here)
  constexpr uint32_t kBatch = 8;
  // Exact evaluation of the records of one batch that pass the screen.  In the cold search no
  // tight bound exists at first (thr32 = +inf: every record passes, and each exact evaluation is
  // a dependent gather from dst), so the record with the smallest screened distance goes first:
  // it almost always is the batch's winner and its bound then rejects the rest.  (A macro, not
  // a lambda: arrays passed by reference end up in scratch memory.)
#define ICP_EXAMINE(t, sc)                                                        \
  do {                                                                            \
    if (COLD && thr32 == __builtin_huge_valf()) {                                 \
      float sm_ = sc[0];                                                          \
      uint32_t ti_ = t[0].idx;                                                    \
      _Pragma("unroll") for (uint32_t u_ = 1; u_ < kBatch; ++u_) {                \
        const bool lt_ = sc[u_] < sm_;                                            \
        sm_ = lt_ ? sc[u_] : sm_;                                                 \
        ti_ = lt_ ? t[u_].idx : ti_;                                              \
      }                                                                           \
      consider(ti_);                                                              \
    }                                                                             \
    _Pragma("unroll") for (uint32_t u_ = 0; u_ < kBatch; ++u_)                    \
      if (!(sc[u_] > thr32)) consider(t[u_].idx);                                 \
  } while (0)
  auto batch = [&](uint32_t p, uint32_t e) {
    NN_STAT(2, 1);
    const uint32_t last = e - 1;
    GridPoint t[kBatch];
#pragma unroll
    for (uint32_t u = 0; u < kBatch; ++u) t[u] = pts[min(p + u, last)];
    float sc[kBatch];
#pragma unroll
    for (uint32_t u = 0; u < kBatch; ++u) sc[u] = screen(t[u]);
    ICP_EXAMINE(t, sc);
  };
  // distance from q to the slab of cells [i0, i1] on axis d (0 inside); outermost cells
  // extend to infinity (targets are clamped into them)
  auto slab = [&](int d, int i0, int i1) -> double {
    const double lo_b = (i0 <= 0) ? -__builtin_huge_val() : (g.lo[d] + i0 * g.h[d]) - mg[d];
    const double hi_b = (i1 >= g.n[d] - 1) ? __builtin_huge_val() : (g.lo[d] + (i1 + 1) * g.h[d]) + mg[d];
    const double v = fmax(lo_b - q[d], q[d] - hi_b);
    return v > 0. ? v : 0.;
  };

  // Walk the rows (iy, iz) of the cell box [lo, hi] in groups of up to four unpruned rows: the
  // bounds of a group are fetched together (one round trip), and its runs are then streamed
  // as ONE flattened sequence, kBatch records in flight -- a typical warm box (2-4 rows of 4-6
  // records) costs one bounds round trip and two record round trips instead of two per row.
  // Rows whose box is strictly farther than the current best cannot win or tie and are skipped
  // (in the warm search `best` starts at the previous match's distance, so the test is already
  // tight when the rows are collected).
  auto walk_box = [&](const int lo[3], const int hi[3]) {
    const double dx = slab(0, lo[0], hi[0]);
    const double dx2 = dx * dx;
    int iz = lo[2], iy = lo[1];
    double dz2 = 0.;
    if (DIM == 3) {
      const double dz = slab(2, iz, iz);
      dz2 = dz * dz;
    }
    unsigned rowno = 0;  // L > 1: rows are dealt round-robin to the lanes of the group
    for (;;) {
      uint32_t ra0 = 0, ra1 = 0, ra2 = 0, ra3 = 0;  // first / one-past-last cell of each collected row
      uint32_t rz0 = 0, rz1 = 0, rz2 = 0, rz3 = 0;
      int nr = 0;
      while (nr < 4 && iz <= hi[2]) {
        if (iy > hi[1]) {
          iy = lo[1];
          ++iz;
          if (DIM == 3 && iz <= hi[2]) {
            const double dz = slab(2, iz, iz);
            dz2 = dz * dz;
          }
          continue;
        }
        const int cy = iy++;
        if (L > 1 && (rowno++ % L) != sub) continue;  // another lane's row
        const double dy = slab(1, cy, cy);
        const double dyz = dy * dy + dz2;
        if (dx2 + dyz > best) continue;
        // a target of this row that can still win or tie has |x - qx| <= sqrt(best - dy^2 - dz^2):
        // clip the run to those cells (the box is a cube, the candidates lie in a ball)
        int xl = lo[0], xh = hi[0];
        if (best < __builtin_huge_val()) {
          const double hw = sqrt(best - dyz) * (1. + 1e-9) + mg[0];
          xl = max(xl, cell_coord(q[0] - hw, g.lo[0], g.inv_h[0], g.n[0]));
          xh = min(xh, cell_coord(q[0] + hw, g.lo[0], g.inv_h[0], g.n[0]));
        }
        const uint32_t rb = ((uint32_t)iz * g.n[1] + cy) * g.n[0];
        const uint32_t ra = rb + xl, rz = rb + xh + 1;
        if (nr == 0) ra0 = ra, rz0 = rz;
        else if (nr == 1) ra1 = ra, rz1 = rz;
        else if (nr == 2) ra2 = ra, rz2 = rz;
        else ra3 = ra, rz3 = rz;
        ++nr;
      }
      if (nr == 0) break;
      NN_STAT(1, nr);
      if (nr < 2) ra1 = ra0, rz1 = rz0;  // unused slots repeat row 0 (a cached address costs next to nothing)
      if (nr < 3) ra2 = ra0, rz2 = rz0;
      if (nr < 4) ra3 = ra0, rz3 = rz0;
      const uint32_t s0 = start[ra0], e0 = start[rz0];
      const uint32_t s1 = start[ra1], e1 = start[rz1];
      const uint32_t s2 = start[ra2], e2 = start[rz2];
      const uint32_t s3 = start[ra3], e3 = start[rz3];
      const uint32_t o1 = e0 - s0;
      const uint32_t o2 = o1 + (nr > 1 ? e1 - s1 : 0u);
      const uint32_t o3 = o2 + (nr > 2 ? e2 - s2 : 0u);
      const uint32_t R = o3 + (nr > 3 ? e3 - s3 : 0u);
      for (uint32_t base = 0; base < R; base += kBatch) {
#ifdef ICP_HACK_CAP
        if (base >= 24) break;
#endif
#ifdef ICP_NN_STATS
        if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) st[4] += 1;
#endif
        NN_STAT(2, 1);
        GridPoint t[kBatch];
#pragma unroll
        for (uint32_t u = 0; u < kBatch; ++u) {
          const uint32_t j = min(base + u, R - 1);  // the tail re-reads the last record
          uint32_t addr = s0 + j;
          if (j >= o1) addr = s1 + (j - o1);
          if (j >= o2) addr = s2 + (j - o2);
          if (j >= o3) addr = s3 + (j - o3);
          t[u] = pts[addr];
        }
        float sc[kBatch];
#pragma unroll
        for (uint32_t u = 0; u < kBatch; ++u) sc[u] = screen(t[u]);
        ICP_EXAMINE(t, sc);
      }
    }
    share_best();  // the group agrees again (nothing below may depend on which lane took which row)
  };

  // Warm start (second and later outer iterations of one estimate call): the pose moved a
  // little, so the previous match is almost always still (nearly) the nearest target.  Its
  // distance bounds the search: every target within sqrt(best) of q lies in a cell of the box
  // [cell(q - r), cell(q + r)] (cell_coord is monotone), typically 1-2 cells per axis instead
  // of the 3^DIM block.  The previous match is a real target and every cell that can hold a
  // closer-or-equal one is visited, so the result is the same exact minimum by (d^2, index).
  bool done = false;
  uint32_t prev_bi = 0xfffffffeu;  // never a target index (icp_create refuses m >= 2^32 - 1)
  if (!COLD && prev) {
    const PrevMatch pm = prev[k];  // coalesced per-slot record (index + exact coordinates), not a gather
    prev_bi = pm.idx;
    if (pm.idx != 0xffffffffu) {
      NN_STAT(7, 1);
      eval(pm.idx, pm.x, pm.y, pm.z);
      if (best < __builtin_huge_val()) {
        const double rad = sqrt(best) * (1. + 1e-9);
        int lo_c[3] = {0, 0, 0}, hi_c[3] = {0, 0, 0};
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          lo_c[d] = cell_coord(q[d] - rad - mg[d], g.lo[d], g.inv_h[d], g.n[d]);
          hi_c[d] = cell_coord(q[d] + rad + mg[d], g.lo[d], g.inv_h[d], g.n[d]);
        }
        walk_box(lo_c, hi_c);
        done = true;
      }
    }
  }

  if (!done) {
    int c[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) c[d] = (d < DIM) ? cell_coord(q[d], g.lo[d], g.inv_h[d], g.n[d]) : 0;
    // Rings 0 and 1 in one go: the block [c-1, c+1]^DIM, centre row first so that `best` is
    // tight early.
    {
      const int x0 = max(c[0] - g.fx, 0), x1 = min(c[0] + g.fx, g.n[0] - 1);
      const uint32_t row = ((uint32_t)c[2] * g.n[1] + c[1]) * g.n[0];
      const uint32_t s = start[row + x0], e = start[row + x1 + 1];
      for (uint32_t p = s; p < e; p += kBatch) batch(p, e);
      int lo_c[3] = {0, 0, 0}, hi_c[3] = {0, 0, 0};
      if (best < __builtin_huge_val()) {
        // the centre row gave a candidate at distance r: every target that can beat or tie it lies
        // in the cell box of the ball, exactly as in the warm search -- one walk instead of shells
        const double rad = sqrt(best) * (1. + 1e-9);
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          lo_c[d] = cell_coord(q[d] - rad - mg[d], g.lo[d], g.inv_h[d], g.n[d]);
          hi_c[d] = cell_coord(q[d] + rad + mg[d], g.lo[d], g.inv_h[d], g.n[d]);
        }
        walk_box(lo_c, hi_c);
        done = true;
      } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const int w = d == 0 ? g.fx : 1;
          lo_c[d] = max(c[d] - w, 0);
          hi_c[d] = min(c[d] + w, g.n[d] - 1);
        }
        walk_box(lo_c, hi_c);  // the other rows of the block are pruned by `best` as it appears
      }
    }
    const int rmax = max(max(g.n[0], g.n[1]), g.n[2]);
    for (int r = 1; r <= rmax && !done; ++r) {
      if (r >= 2) {  // shell r of the general walk (rings 0 and 1 were handled above)
        const int z0 = DIM == 3 ? max(c[2] - r, 0) : 0, z1 = DIM == 3 ? min(c[2] + r, g.n[2] - 1) : 0;
        const int y0 = max(c[1] - r, 0), y1 = min(c[1] + r, g.n[1] - 1);
        for (int iz = z0; iz <= z1; ++iz) {
          const bool ze = DIM == 3 && (iz == c[2] - r || iz == c[2] + r);
          const double dz = DIM == 3 ? slab(2, iz, iz) : 0.;
          for (int iy = y0; iy <= y1; ++iy) {
            const bool edge = ze || iy == c[1] - r || iy == c[1] + r;
            const double dy = slab(1, iy, iy);
            const double dyz = dy * dy + dz * dz;
            if (dyz > best) continue;
            // cells of this row that belong to shell r: the whole run on an edge row, else
            // only the two end cells
            const int nruns = edge ? 1 : 2;
            for (int run = 0; run < nruns; ++run) {
              int x0, x1;
              if (edge) {
                x0 = c[0] - r * g.fx;
                x1 = c[0] + r * g.fx;
              } else if (run == 0) {
                x0 = c[0] - r * g.fx;
                x1 = c[0] - (r - 1) * g.fx - 1;
              } else {
                x0 = c[0] + (r - 1) * g.fx + 1;
                x1 = c[0] + r * g.fx;
              }
              if (x1 < 0 || x0 > g.n[0] - 1) continue;
              x0 = max(x0, 0);
              x1 = min(x1, g.n[0] - 1);
              const double dx = slab(0, x0, x1);
              if (dx * dx + dyz > best) continue;  // strictly farther: cannot win or tie
              const uint32_t row = ((uint32_t)iz * g.n[1] + iy) * g.n[0];
              const uint32_t s = start[row + x0], e = start[row + x1 + 1];
              for (uint32_t p = s; p < e; p += kBatch) batch(p, e);
            }
          }
        }
      }
      // can anything outside the visited block [c-r, c+r] (x: r * fx cells) still win or tie?
      double Lb = __builtin_huge_val();
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        const int w = d == 0 ? r * g.fx : r;
        if (c[d] - w > 0) Lb = fmin(Lb, (q[d] - (g.lo[d] + (c[d] - w) * g.h[d])) - mg[d]);
        if (c[d] + w < g.n[d] - 1) Lb = fmin(Lb, ((g.lo[d] + (c[d] + w + 1) * g.h[d]) - q[d]) - mg[d]);
      }
      if (Lb == __builtin_huge_val()) break;  // the whole grid has been visited
      if (Lb > 0. && best < Lb * Lb) break;   // every unvisited target is strictly farther
    }
  }
#ifdef ICP_NN_STATS
  {
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < 5; ++j)
      if (j != 4) atomicAdd(&g_nn_stats[j], (unsigned long long)st[j]);
    atomicAdd(&g_nn_stats[4], (unsigned long long)st[4]);
    atomicAdd(&g_nn_stats[7], (unsigned long long)st[7]);
    atomicAdd(&g_nn_hist[0][st[2] < 31 ? st[2] : 31], 1ull);
    {
      unsigned mx = st[2];
      for (int off = 32; off >= 1; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off));
      if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) atomicAdd(&g_nn_hist[1][mx < 31 ? mx : 31], 1ull);
    }
    if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) {
      atomicAdd(&g_nn_stats[5], t_end - t_begin);
      atomicAdd(&g_nn_stats[6], 1ull);
    }
  }
#endif
  // prev and prev_out are the same per-slot array: a slot whose match did not change already
  // holds this record (32 B of write traffic per query saved once the registration settles)
  if (L > 1 && sub != 0) return;  // the group agrees (walk_box ends on share_best); one lane reports
  if (prev_out && bi != prev_bi) {
    PrevMatch pm;
    pm.x = bx;
    pm.y = by;
    pm.z = bz;
    pm.idx = bi;
    pm.pad = 0;
    prev_out[k] = pm;
  }
  if (bi == 0xffffffffu) {  // no finite distance at all (NaN query): index 0, as a scan from 0 would
    bi = 0;
    bx = dst[0];
    by = dst[1];
  }
  if (idx) idx[i] = bi;
  if (a) a[i] = make_double2(q[0], q[1]);
  if (b) b[i] = make_double2(bx, by);
}

// ---------------------------------------------------------------- warm search ----
// The search of the second and later outer iterations of one estimate call, one lane per query
// (clouds beyond ICP_NN_COOP_MAX_N points), rewritten in round 2 after the counters showed what the
// kernel above is bound by: NOT memory latency but vector-instruction issue -- 2 600 VALU
// instructions per wave, ~80 % of the SIMDs' issue slots (profiles/r02_nn_grid_sq_pmc.txt); f64
// square roots (a 20-instruction sequence each: radius, per-row clip, every improvement), f64 floor /
// clamp chains for every cell coordinate, a 12-instruction select chain per record to flatten the
// runs of a row group.  Same search, same result -- the winner is still decided by the contract's
// exact f64 distance and (d^2, index) order, so the indices are bit-identical to the sweep's -- with
// everything that only PRUNES evaluated in f32 relative to the grid origin:
//   * every bound is conservative by construction: radii are rounded up, distances to cell slabs
//     down, by margins `mgf` / `em` that dominate the f32 rounding of the quantities involved
//     (derivations at the definitions); a bound can therefore only cause extra visits;
//   * the records of a row are read in QUADS of four (64 bytes): a run [s, e) becomes the quads that
//     start at s, s + 4, ... (round 6; rounds 2-5 aligned them to four records: up to three extra
//     records on EITHER side of the run, a quarter more quads for the usual runs of three to six), so
//     the flattening arithmetic is paid per quad, not per record.  The extra records the last quad
//     drags in are real targets of the cells behind the run (or the +inf sentinels behind the last
//     record): screening them can add candidates, never remove one.
// Lanes whose geometry does not fit f32 (|q - lo| or the radius beyond 1e18) walk the whole grid
// unpruned: correct, and never seen outside adversarial tests.
#ifdef ICP_WARM_WAVES
#define ICP_WARM_ATTR __attribute__((amdgpu_waves_per_eu(ICP_WARM_WAVES, ICP_WARM_WAVES)))
#else
#define ICP_WARM_ATTR
#endif
#ifndef ICP_WARM_QUADS
#define ICP_WARM_QUADS 2  // quads (of four records) in flight per lane
#endif
// CERT: the walk also leaves a CERTIFICATE in the slot's record (PrevMatch::pad, see k_nn_cert below): a lower
// bound on the distance from this query to every target other than its match, from what the walk saw anyway --
// the second smallest screened distance among the records it visited, and the distance to everything it skipped
// (rows beyond the radius, cells clipped off a row, the cells outside its box), each bound rounded down by the
// margins that already make the pruning conservative.
struct CertDecay {  // how far a query can have moved since the snapshot was taken, per unit of |s_xy| and flat:
  float r_lo, t_lo;  // lower bounds (a certificate is stored relative to the snapshot: + decay at creation)
  float r_hi, t_hi;  // upper bounds (... and checked against the decay at the time of the check)
};

// SEEDED: the previous match comes in registers (`seed`: the first search of a snapshot, k_nn_grid_seeded) and the
// slot's record is written whatever the walk finds.
template <int DIM, bool CERT = false, bool SEEDED = false>
__device__ __forceinline__ void warm_query(const unsigned k, const double *__restrict__ src,
                                           const uint32_t *__restrict__ perm, Pose T, const GridParams &g,
                                           const uint32_t *__restrict__ start, const GridPoint *__restrict__ pts,
                                           const double *__restrict__ dst, uint32_t *__restrict__ idx,
                                           double2 *__restrict__ a, double2 *__restrict__ b, PrevMatch *prev,
                                           CertDecay cd = CertDecay{0.f, 0.f, 0.f, 0.f}, PrevMatch seed = PrevMatch{0., 0., 0., 0xffffffffu, 0u}) {
  const unsigned i = perm ? perm[k] : k;  // null: outputs in slot order
#ifdef ICP_NN_STATS
  unsigned ws[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long wt0 = __builtin_amdgcn_s_memtime();
#define WARM_LEADER() ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)__ballot(1)) - 1))
#endif
  double q[3];
  q[0] = src[(size_t)k * DIM + 0];
  q[1] = src[(size_t)k * DIM + 1];
  q[2] = DIM == 3 ? src[(size_t)k * DIM + 2] : 0.;
  float s_norm_lo = 0.f;  // CERT: a lower bound of |s_xy|
  if (CERT) s_norm_lo = __builtin_amdgcn_sqrtf(fmaxf((float)(q[0] * q[0] + q[1] * q[1]) * 0.9999997f, 0.f)) * 0.9999997f;
  {  // Transform::transform, src/transform.rs:22-24
    const double nx = (T.r00 * q[0] + T.r01 * q[1]) + T.tx;
    const double ny = (T.r10 * q[0] + T.r11 * q[1]) + T.ty;
    q[0] = nx;
    q[1] = ny;
  }
  const PrevMatch pm = SEEDED ? seed : prev[k];
  if (pm.idx == 0xffffffffu) {  // no finite distance was ever found (NaN query): index 0, as a scan from 0 would
    if (SEEDED) prev[k] = pm;
    if (idx) idx[i] = 0;
    if (a) a[i] = make_double2(q[0], q[1]);
    if (b) b[i] = make_double2(dst[0], dst[1]);
    return;
  }
  // the contract's exact distance: d^2 = ((dx*dx + dy*dy) + dz*dz), no FMA
  auto dist2 = [&](double tx, double ty, double tz) -> double {
    const double ddx = q[0] - tx, ddy = q[1] - ty;
    double dd = ddx * ddx + ddy * ddy;
    if (DIM == 3) {
      const double ddz = q[2] - tz;
      dd = dd + ddz * ddz;
    }
    return dd;
  };
  double best = dist2(pm.x, pm.y, pm.z);
  uint32_t bi = pm.idx;
  double bx = pm.x, by = pm.y, bz = pm.z;
  if (!(best == best)) {  // a NaN distance compares false with everything: start without a match instead
    best = __builtin_huge_val();
    bi = 0xffffffffu;
  }

  // ---- f32 geometry relative to the grid origin ----
  float qf[3], amax = 0.f;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    qf[d] = d < DIM ? (float)(q[d] - g.lo[d]) : 0.f;  // |error| <= 2^-24 |q - lo|
    amax = fmaxf(amax, fabsf(qf[d]));
  }
  // mgf: absolute margin of every f32 length below.  A grid-relative coordinate (|.| <= amax), a cell
  // edge i * hf (|.| <= g.ext) and a handful of additions / multiplications of them are each within
  // 2^-24 relative of the exact value: 4e-7 (amax + ext) covers six such roundings.
  const float mgf = 4e-7f * (amax + g.ext);
  // ecf: the screen's bound on |(qf - pf) - (q - p)| (three f32 roundings per component, sqrt(3) for
  // the norm): 2^-23 (ext + |q - lo|) sqrt(3) = 2.07e-7 (...)
  const float ecf = 2.1e-7f * (amax + g.ext);
  float bf, rf, thr32;  // bf >= best; rf >= sqrt(best) + mgf; thr32: records with s32 > thr32 cannot win or tie
  auto set_radius = [&]() {
    bf = fmaxf((float)best * 1.0000003f, 1e-37f);      // (float) rounds to nearest: the factor restores >=
    const float rs = __builtin_amdgcn_sqrtf(bf) * 1.0000003f;  // >= sqrt(best) (v_sqrt_f32: 1 ulp)
    rf = rs + mgf;
    // a target within sqrt(best) has |qf - pf| <= sqrt(best) + ecf, and its f32-evaluated square is
    // at most 1.1e-6 above the exact one (nn_grid.hip, top of k_nn_grid)
    thr32 = (rs + ecf) * (rs + ecf) * 1.000005f;
  };
  set_radius();
  const bool wide = !(amax + rf < 1e18f);  // (also NaN): no f32 geometry for this lane
  if (wide) {
    bf = __builtin_huge_valf();
    thr32 = __builtin_huge_valf();
  }
  const float hf[3] = {(float)g.h[0], (float)g.h[1], (float)g.h[2]};
  const float ihf[3] = {(float)g.inv_h[0], (float)g.inv_h[1], (float)g.inv_h[2]};
  // cell of a grid-relative coordinate, rounded towards `dir` by more than the f32 error of the
  // product: |t - exact| <= 3e-7 (|v| inv_h); targets were binned in f64 (k_grid_count), whose own
  // rounding (1e-12 cells) hides in the constant
  auto cell_lo = [&](float v, float em, int d) -> int {
    const float t = fminf(fmaxf(__builtin_floorf(v * ihf[d] - em), 0.f), (float)(g.n[d] - 1));
    return (int)t;
  };
  auto cell_hi = [&](float v, float em, int d) -> int {
    const float t = fminf(fmaxf(__builtin_floorf(v * ihf[d] + em), 0.f), (float)(g.n[d] - 1));
    return (int)t;
  };
  int lo_c[3] = {0, 0, 0}, hi_c[3] = {0, 0, 0};
  float em[3];
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    em[d] = (fabsf(qf[d]) + rf) * ihf[d] * 4e-7f + 1e-3f;
    lo_c[d] = wide ? 0 : cell_lo(qf[d] - rf, em[d], d);
    hi_c[d] = wide ? g.n[d] - 1 : cell_hi(qf[d] + rf, em[d], d);
  }
  // squared distance from q to the slab of cells [c, c] on axis d, rounded DOWN (0 inside; the
  // outermost cells extend to infinity: targets are clamped into them)
  auto slab2 = [&](int d, int c) -> float {
    const float e0 = (float)c * hf[d];
    const float below = c <= 0 ? -__builtin_huge_valf() : e0 - qf[d];
    const float above = c >= g.n[d] - 1 ? -__builtin_huge_valf() : qf[d] - (e0 + hf[d]);
    const float v = fmaxf(fmaxf(below, above) - mgf, 0.f);
    return v * v * 0.9999997f;
  };
  // CERT: m2 = the smallest screened squared distance among the visited records other than the match (a match that
  // is replaced joins with its exact distance); skip2 = the smallest squared distance bound of anything not visited
  float m2 = __builtin_huge_valf(), skip2 = __builtin_huge_valf();
  if (CERT && !wide) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) {  // targets in cells outside the box [lo_c, hi_c] (slab2's argument, for whole half-spaces)
      if (lo_c[d] > 0) {
        const float gq = fmaxf(qf[d] - (float)lo_c[d] * hf[d] - mgf, 0.f);
        skip2 = fminf(skip2, gq * gq * 0.9999997f);
      }
      if (hi_c[d] < g.n[d] - 1) {
        const float gq = fmaxf((float)(hi_c[d] + 1) * hf[d] - qf[d] - mgf, 0.f);
        skip2 = fminf(skip2, gq * gq * 0.9999997f);
      }
    }
  }
  auto consider = [&](uint32_t ti) {
    const double tx = dst[(size_t)ti * DIM + 0], ty = dst[(size_t)ti * DIM + 1];
    const double tz = DIM == 3 ? dst[(size_t)ti * DIM + 2] : 0.;
    const double dd = dist2(tx, ty, tz);
    if (dd < best || (dd == best && ti < bi)) {
      if (CERT && bi != 0xffffffffu) m2 = fminf(m2, (float)best * 0.9999997f);  // the match it replaces is "another target" now
      best = dd;
      bi = ti;
      bx = tx;
      by = ty;
      bz = tz;
      if (!wide) set_radius();
    }
  };

  int iz = lo_c[2], iy = lo_c[1];
  float dz2 = (DIM == 3 && !wide) ? slab2(2, iz) : 0.f;
  for (;;) {
    // up to four unpruned rows: first / one-past-last QUAD of their (clipped) runs' cells
    uint32_t ra0 = 0, ra1 = 0, ra2 = 0, ra3 = 0, rz0 = 0, rz1 = 0, rz2 = 0, rz3 = 0;
    int nr = 0;
    while (nr < 4 && iz <= hi_c[2]) {
      if (iy > hi_c[1]) {
        iy = lo_c[1];
        ++iz;
        if (DIM == 3 && !wide && iz <= hi_c[2]) dz2 = slab2(2, iz);
        continue;
      }
      const int cy = iy++;
      int xl = lo_c[0], xh = hi_c[0];
      if (!wide) {
        const float dyz = slab2(1, cy) + dz2;
        if (dyz > bf) {  // every target of the row is strictly farther than the best so far
          if (CERT) skip2 = fminf(skip2, dyz);
          continue;
        }
        // a target of this row that can still win or tie has |x - qx| <= sqrt(best - dy^2 - dz^2)
        // (v_sqrt_f32 returns 0 for a denormal argument: sqrt of it is < 1.1e-19 <= mgf, build_grid's f32_ok)
        const float hw = __builtin_amdgcn_sqrtf(bf - dyz) * 1.000001f + mgf;
        xl = max(xl, cell_lo(qf[0] - hw, em[0], 0));
        xh = min(xh, cell_hi(qf[0] + hw, em[0], 0));
        if (xl > xh) {
          if (CERT) skip2 = 0.f;  // (cannot happen: the query's own cell is in both ranges; no certificate if it does)
          continue;
        }
        if (CERT) {  // the cells of the row on either side of [xl, xh]
          if (xl > lo_c[0]) skip2 = fminf(skip2, slab2(0, xl - 1) + dyz);
          if (xh < hi_c[0]) skip2 = fminf(skip2, slab2(0, xh + 1) + dyz);
        }
      }
      const uint32_t rb = ((uint32_t)iz * g.n[1] + cy) * g.n[0];
      const uint32_t ra = rb + xl, rz = rb + xh + 1;
      if (nr == 0) ra0 = ra, rz0 = rz;
      else if (nr == 1) ra1 = ra, rz1 = rz;
      else if (nr == 2) ra2 = ra, rz2 = rz;
      else ra3 = ra, rz3 = rz;
      ++nr;
    }
    if (nr == 0) break;
#ifdef ICP_NN_STATS
    ws[3] += 1;
    if (WARM_LEADER()) ws[0] += 1;
#endif
    if (nr < 2) ra1 = ra0, rz1 = rz0;  // unused slots repeat row 0 (a cached address)
    if (nr < 3) ra2 = ra0, rz2 = rz0;
    if (nr < 4) ra3 = ra0, rz3 = rz0;
    const uint32_t s0 = start[ra0], e0 = start[rz0];
    const uint32_t s1 = start[ra1], e1 = start[rz1];
    const uint32_t s2 = start[ra2], e2 = start[rz2];
    const uint32_t s3 = start[ra3], e3 = start[rz3];
    // runs -> quads that start at the run's first record (round 6: see warm_wave); an empty run has no quads
    const uint32_t n0 = e0 > s0 ? (e0 - s0 + 3) >> 2 : 0u;
    const uint32_t n1 = (nr > 1 && e1 > s1) ? (e1 - s1 + 3) >> 2 : 0u;
    const uint32_t n2 = (nr > 2 && e2 > s2) ? (e2 - s2 + 3) >> 2 : 0u;
    const uint32_t n3 = (nr > 3 && e3 > s3) ? (e3 - s3 + 3) >> 2 : 0u;
    const uint32_t o1 = n0, o2 = o1 + n1, o3 = o2 + n2, Q = o3 + n3;
    // quad j of the flattened sequence starts at record 4 j + dk of the record array (modulo 2^32)
    const uint32_t d0 = s0, d1 = s1 - 4u * o1, d2 = s2 - 4u * o2, d3 = s3 - 4u * o3;
    constexpr uint32_t kQ = ICP_WARM_QUADS, kR = 4 * kQ;
    for (uint32_t base = 0; base < Q; base += kQ) {
      GridPoint t[kR];
      const uint4 *line[kQ];
#pragma unroll
      for (uint32_t h2 = 0; h2 < kQ; ++h2) {
        const uint32_t j = min(base + h2, Q - 1);  // the tail re-reads the last quad
        uint32_t dq = d0;
        if (j >= o1) dq = d1;
        if (j >= o2) dq = d2;
        if (j >= o3) dq = d3;
        line[h2] = reinterpret_cast<const uint4 *>(pts) + (size_t)(uint32_t)(4u * j + dq);
      }
#pragma unroll
      for (uint32_t u = 0; u < kR; ++u) {  // all loads in flight before the first use
        const uint4 w = line[u >> 2][u & 3];
        t[u].x = __uint_as_float(w.x);
        t[u].y = __uint_as_float(w.y);
        t[u].z = __uint_as_float(w.z);
        t[u].idx = w.w;
      }
      __builtin_amdgcn_sched_barrier(0);
      float sc[kR];
#pragma unroll
      for (uint32_t u = 0; u < kR; ++u) {
        const float fx = qf[0] - t[u].x, fy = qf[1] - t[u].y;
        float s2 = __builtin_fmaf(fy, fy, fx * fx);
        if (DIM == 3) {
          const float fz = qf[2] - t[u].z;
          s2 = __builtin_fmaf(fz, fz, s2);
        }
        sc[u] = s2;
      }
#ifdef ICP_NN_STATS
      {
        bool any = false;
        for (uint32_t u = 0; u < kR; ++u)
          if (!(sc[u] > thr32) && t[u].idx != bi) any = true, ws[5] += 1;
        ws[4] += 1;
        const bool wave_any = __ballot(any) != 0ull;
        if (WARM_LEADER()) ws[1] += 1, ws[2] += wave_any ? 1u : 0u;
      }
#endif
#pragma unroll
      for (uint32_t u = 0; u < kR; ++u)
        if (!(sc[u] > thr32) && t[u].idx != bi) consider(t[u].idx);
      if (CERT) {
#pragma unroll
        for (uint32_t u = 0; u < kR; ++u) m2 = fminf(m2, t[u].idx != bi ? sc[u] : __builtin_huge_valf());
      }
    }
  }
#ifdef ICP_NN_STATS
  {
    // (lanes leave the loops at different times: the wave-level counts belong to whichever lane led at the time)
    unsigned w0 = ws[0], w1 = ws[1], w2 = ws[2];
    for (int off = 32; off >= 1; off >>= 1) {
      w0 += (unsigned)__shfl_xor((int)w0, off);
      w1 += (unsigned)__shfl_xor((int)w1, off);
      w2 += (unsigned)__shfl_xor((int)w2, off);
    }
    atomicAdd(&g_nn_warm[3], (unsigned long long)ws[3]);
    atomicAdd(&g_nn_warm[4], (unsigned long long)ws[4]);
    atomicAdd(&g_nn_warm[5], (unsigned long long)ws[5]);
    if (WARM_LEADER()) {
      atomicAdd(&g_nn_warm[0], (unsigned long long)w0);
      atomicAdd(&g_nn_warm[1], (unsigned long long)w1);
      atomicAdd(&g_nn_warm[2], (unsigned long long)w2);
      atomicAdd(&g_nn_warm[6], __builtin_amdgcn_s_memtime() - wt0);
      atomicAdd(&g_nn_warm[7], 1ull);
      atomicAdd(&g_nn_warm_hist[0][w0 < 31 ? w0 : 31], 1ull);
      atomicAdd(&g_nn_warm_hist[1][w1 < 31 ? w1 : 31], 1ull);
    }
  }
#undef WARM_LEADER
#endif
  uint32_t cert_bits = 0;
  if (CERT && !wide && bi != 0xffffffffu) {
    // distances from squared bounds: a screened square is within 4e-7 relative of |qf - pf|^2, and |q - p| is
    // within ecf of |qf - pf| (the screen's own error budget above); the skip bounds are lower bounds already
    const float seen = __builtin_amdgcn_sqrtf(m2) * 0.9999994f - ecf;
    const float skipped = __builtin_amdgcn_sqrtf(skip2) * 0.9999997f;
    const float cert = fminf(seen, skipped);
    const float abs_lo = (cert + (cd.r_lo * s_norm_lo + cd.t_lo) * 0.9999997f) * 0.9999997f;
    if (cert > 0.f && abs_lo > 0.f && abs_lo < __builtin_huge_valf()) cert_bits = __float_as_uint(abs_lo);
  }
  if (bi == 0xffffffffu) {  // no finite distance at all: index 0, as a scan from 0 would (k_nn_grid does the same)
    bi = 0;
    bx = dst[0];
    by = dst[1];
    bz = DIM == 3 ? dst[2] : 0.;
  }
  // a slot whose match did not change already holds this record
  if (CERT || SEEDED || bi != pm.idx) {
    PrevMatch out;
    out.x = bx;
    out.y = by;
    out.z = bz;
    out.idx = bi;
    out.pad = cert_bits;
    prev[k] = out;
  }
  if (idx) idx[i] = bi;
  if (a) a[i] = make_double2(q[0], q[1]);
  if (b) b[i] = make_double2(bx, by);
}

template <int DIM, bool CERT>
__global__ __launch_bounds__(kGridThreads) ICP_WARM_ATTR void k_nn_grid_warm(const double *__restrict__ src,
                                                               const uint32_t *__restrict__ perm, unsigned n, Pose T,
                                                               GridParams g, const uint32_t *__restrict__ start,
                                                               const GridPoint *__restrict__ pts,
                                                               const double *__restrict__ dst, uint32_t *__restrict__ idx,
                                                               double2 *__restrict__ a, double2 *__restrict__ b,
                                                               PrevMatch *prev, CertDecay cd, unsigned xcd_chunk) {
  const unsigned k = xcd_wave(blockIdx.x, gridDim.x, xcd_chunk) * kGridThreads + threadIdx.x;
  if (k >= n) return;
  warm_query<DIM, CERT>(k, src, perm, T, g, start, pts, dst, idx, a, b, prev, cd);
}

// ------------------------------------------------------ the warm walk, shared by the wave (round 4) ----
// In warm_query a lane walks its own query, and a wave lasts as long as its slowest lane: on the benchmark pair a lane
// visits 2.4 chunks of records on average and its WAVE 8 -- thirteen dependent trips to memory per wave, two thirds of
// the record slots it loads wasted on lanes that had finished (profiles/r04_warm_walk_stats.txt).  Here the lanes of a
// wave pool their work.  Per round every lane, as OWNER of its query, picks its next (up to four) rows exactly as
// warm_query does and fetches their bounds; the quads of all 64 owners form one flat list (a prefix sum over the lanes);
// every lane, as WORKER, takes the quads lane, lane + 64, ... of that list whoever they belong to -- finds the owner by a
// binary search in the prefix sums, screens the four records against the owner's query and radius (LDS) -- and records
// that pass go to a candidate list.  The exact tests of the list run together (flush: one gather of f64 coordinates for
// the whole wave instead of one per lane and chunk), each by whichever lane holds the entry, with the owner's f64 query
// fetched by ds_bpermute and the SAME operations as warm_query's dist2; the owners then take the lexicographic minimum
// (d^2, index) over their entries.  That minimum does not depend on the order of the candidates, the rows an owner
// visits are those warm_query would visit with a radius at least as large (the radius is refreshed only at a flush), so
// the result is the same index -- tests/test_gpu_parity.py compares the two kernels on every cloud it has.
#ifndef ICP_COOP_ITEMS
#define ICP_COOP_ITEMS 2
#endif
#ifndef ICP_COOP_ROWS
#define ICP_COOP_ROWS 4
#endif
#ifndef ICP_COOP_FLUSH_AT
#define ICP_COOP_FLUSH_AT 96
#endif
constexpr int kCoopItems = ICP_COOP_ITEMS;  // quads in flight per worker lane
constexpr int kCoopRows = ICP_COOP_ROWS;    // rows an owner contributes per round (4 or 8)
constexpr int kCoopCandCap = 256;           // >= 4 records x 64 lanes: one quad per lane always fits after a flush
constexpr int kCoopFlushAt = ICP_COOP_FLUSH_AT;  // pending exact tests that are worth a gather between two rounds of rows
struct CoopLds {
  uint32_t pref[64];            // quads of the owners before this one
  uint32_t rows_d[64][kCoopRows];  // quad j of owner's flattened rows starts at record 4 j + d_r ...
  uint32_t rows_o[64][kCoopRows];  // ... r = the number of o_1.. that j has reached (o_0 unused)
  float4 oq[64];                // the owner's grid-relative query and screening threshold
  uint32_t obi[64];             // the owner's current match
  uint2 cand[kCoopCandCap];     // (owner, target index)
  unsigned long long fkey[64];  // flush: the smallest d^2 (as bits) among an owner's entries, ...
  uint32_t fidx[64];            // ... the lowest index among those, ...
  uint32_t fwin[64];            // ... and the lane that holds that entry's coordinates
};

#ifdef ICP_COOP_PROFILE
// diagnostic build: per-phase time of a wave (10 ns ticks, summed over waves; slot = workgroup % 64)
// [0] query + previous match loaded, geometry  [1] row selection  [2] row bounds (start[]) arrived  [3] prefix + tables
// [4] worker rounds  [5] flushes  [6] outputs  [7] waves  [8] rounds of rows  [9] worker rounds  [10] flushes  [11] candidates
__device__ unsigned long long g_coop_prof[64][12];
__device__ long long g_coop_span[16384][2];  // start / end tick of every wave (workgroup) of the last launch
__device__ unsigned long long g_coop_rows[32];  // owners by the number of box rows their walk took (31: more); profiles/coop_rows_hist.py
#define COOP_STAMP(i)                                                  \
  do {                                                                 \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        \
    const long long now_ = wall_clock64();                             \
    cprof[i] += (unsigned long long)(now_ - ct_last);                  \
    ct_last = now_;                                                    \
  } while (0)
#define COOP_COUNT(i, v) (cprof[i] += (v))
#else
#define COOP_STAMP(i) ((void)0)
#define COOP_COUNT(i, v) ((void)0)
#endif

template <int DIM, bool SEEDED>
__device__ __forceinline__ void warm_wave(const unsigned k, const unsigned n, const double *__restrict__ src,
                                          const uint32_t *__restrict__ perm, Pose T, const GridParams &g,
                                          const uint32_t *__restrict__ start, const GridPoint *__restrict__ pts,
                                          const double *__restrict__ dst, uint32_t *__restrict__ idx,
                                          double2 *__restrict__ a, double2 *__restrict__ b, PrevMatch *prev, PrevMatch seed,
                                          CoopLds &S) {
  const unsigned lane = threadIdx.x;  // one wave per workgroup
#ifdef ICP_COOP_PROFILE
  unsigned long long cprof[12] = {0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0};
  long long ct_last = wall_clock64();
  const long long ct_first = ct_last;
#endif
  const bool in_range = k < n;
  const unsigned kk = in_range ? k : n - 1;  // the lanes past the end repeat the last query and store nothing
  const unsigned i = perm ? perm[kk] : kk;
  double q[3];
  q[0] = src[(size_t)kk * DIM + 0];
  q[1] = src[(size_t)kk * DIM + 1];
  q[2] = DIM == 3 ? src[(size_t)kk * DIM + 2] : 0.;
  {  // Transform::transform, src/transform.rs:22-24
    const double nx = (T.r00 * q[0] + T.r01 * q[1]) + T.tx;
    const double ny = (T.r10 * q[0] + T.r11 * q[1]) + T.ty;
    q[0] = nx;
    q[1] = ny;
  }
  const PrevMatch pm = SEEDED ? seed : prev[kk];
  const bool no_match = pm.idx == 0xffffffffu;  // no finite distance was ever found (NaN query): index 0, no walk
  bool alive = in_range && !no_match;
  auto dist2 = [&](double tx, double ty, double tz) -> double {  // the contract's exact distance (warm_query)
    const double ddx = q[0] - tx, ddy = q[1] - ty;
    double dd = ddx * ddx + ddy * ddy;
    if (DIM == 3) {
      const double ddz = q[2] - tz;
      dd = dd + ddz * ddz;
    }
    return dd;
  };
  double best = dist2(pm.x, pm.y, pm.z);
  uint32_t bi = pm.idx;
  double bx = pm.x, by = pm.y, bz = pm.z;
  if (!(best == best)) {
    best = __builtin_huge_val();
    bi = 0xffffffffu;
  }
  // ---- f32 geometry relative to the grid origin: warm_query's, margin for margin ----
  float qf[3], amax = 0.f;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    qf[d] = d < DIM ? (float)(q[d] - g.lo[d]) : 0.f;
    amax = fmaxf(amax, fabsf(qf[d]));
  }
  const float mgf = 4e-7f * (amax + g.ext);
  const float ecf = 2.1e-7f * (amax + g.ext);
  float bf, rf, thr32;
  auto set_radius = [&]() {
    bf = fmaxf((float)best * 1.0000003f, 1e-37f);
    const float rs = __builtin_amdgcn_sqrtf(bf) * 1.0000003f;
    rf = rs + mgf;
    thr32 = (rs + ecf) * (rs + ecf) * 1.000005f;
  };
  set_radius();
  const bool wide = !(amax + rf < 1e18f);
  if (wide) {
    bf = __builtin_huge_valf();
    thr32 = __builtin_huge_valf();
  }
  const float hf[3] = {g.hf[0], g.hf[1], g.hf[2]};
  const float ihf[3] = {g.ihf[0], g.ihf[1], g.ihf[2]};
  auto cell_lo = [&](float v, float em, int d) -> int {
    const float t = fminf(fmaxf(__builtin_floorf(v * ihf[d] - em), 0.f), g.nm1f[d]);
    return (int)t;
  };
  auto cell_hi = [&](float v, float em, int d) -> int {
    const float t = fminf(fmaxf(__builtin_floorf(v * ihf[d] + em), 0.f), g.nm1f[d]);
    return (int)t;
  };
  int lo_c[3] = {0, 0, 0}, hi_c[3] = {0, 0, 0};
  float em[3];
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    em[d] = (fabsf(qf[d]) + rf) * ihf[d] * 4e-7f + 1e-3f;
    lo_c[d] = wide ? 0 : cell_lo(qf[d] - rf, em[d], d);
    hi_c[d] = wide ? g.n[d] - 1 : cell_hi(qf[d] + rf, em[d], d);
  }
  auto slab2 = [&](int d, int c) -> float {
    const float e0 = (float)c * hf[d];
    const float below = c <= 0 ? -__builtin_huge_valf() : e0 - qf[d];
    const float above = c >= g.n[d] - 1 ? -__builtin_huge_valf() : qf[d] - (e0 + hf[d]);
    const float v = fmaxf(fmaxf(below, above) - mgf, 0.f);
    return v * v * 0.9999997f;
  };
  S.oq[lane] = make_float4(qf[0], qf[1], qf[2], thr32);
  S.obi[lane] = bi;
  S.fkey[lane] = 0x7ff0000000000000ull;
  S.fidx[lane] = 0xffffffffu;
  S.fwin[lane] = lane;
  COOP_STAMP(0);

  unsigned cnt = 0;  // pending candidates: the same number in every lane (ballots)
  // the exact tests of the pending candidates, merged into their owners' matches
  auto flush = [&]() {
    COOP_COUNT(10, 1);
    COOP_COUNT(11, cnt);
    for (unsigned base = 0; base < cnt; base += 64) {
      const unsigned e = base + lane;
      const bool have = e < cnt;
      const uint2 c = S.cand[have ? e : base];
      const uint32_t ti = c.y;
      const double tx = dst[(size_t)ti * DIM + 0], ty = dst[(size_t)ti * DIM + 1];
      const double tz = DIM == 3 ? dst[(size_t)ti * DIM + 2] : 0.;
      // the owner's query, bit for bit
      const int sel = (int)(c.x << 2);
      double oqd[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(sel, (int)__double2loint(q[d]));
        const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(sel, (int)__double2hiint(q[d]));
        oqd[d] = __hiloint2double((int)hi, (int)lo);
      }
      const double ddx = oqd[0] - tx, ddy = oqd[1] - ty;
      double dd = ddx * ddx + ddy * ddy;
      if (DIM == 3) {
        const double ddz = oqd[2] - tz;
        dd = dd + ddz * ddz;
      }
      // every owner's lexicographic minimum (d^2, index) over the entries of this batch, by LDS atomics: d^2 >= 0 or
      // NaN, so its bit pattern orders like the value and a NaN sorts behind +inf (never chosen, as in warm_query)
      const unsigned long long key = (unsigned long long)__double_as_longlong(dd);
      if (have) atomicMin(&S.fkey[c.x], key);
      __syncthreads();
      const bool first = have && S.fkey[c.x] == key;
      if (first) atomicMin(&S.fidx[c.x], ti);
      __syncthreads();
      if (first && S.fidx[c.x] == ti) S.fwin[c.x] = lane;  // (the same target twice in the list: either lane, same values)
      __syncthreads();
      const double d2 = __longlong_as_double((long long)S.fkey[lane]);
      const uint32_t t2 = S.fidx[lane];
      const int wsel = (int)(S.fwin[lane] << 2);
      double wx[3];
      {
        const double tc[3] = {tx, ty, tz};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(wsel, (int)__double2loint(tc[d]));
          const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(wsel, (int)__double2hiint(tc[d]));
          wx[d] = __hiloint2double((int)hi, (int)lo);
        }
      }
      if (d2 < best || (d2 == best && t2 < bi)) {
        best = d2;
        bi = t2;
        bx = wx[0];
        by = wx[1];
        bz = wx[2];
      }
      S.fkey[lane] = 0x7ff0000000000000ull;  // + infinity
      S.fidx[lane] = 0xffffffffu;
      S.fwin[lane] = lane;
      __syncthreads();
    }
    cnt = 0;
    if (!wide) set_radius();
    __syncthreads();  // (the workers' reads of oq / obi / cand are behind us)
    S.oq[lane].w = thr32;
    S.obi[lane] = bi;
    __syncthreads();
  };

  // The rows of the owner's box, a TILE of 4 x 4 (y x z) at a time: one bit per row that can hold a winner (warm_query's
  // test: the row's slab distance against the radius), computed for all sixteen at once -- straight-line code that every
  // lane runs once for a box of up to 4 x 4 rows, where warm_query's loop over the rows runs as long as the wave's largest
  // box (half of the vector instructions of the first version of this kernel: profiles/r04_search_coop.txt).  A row
  // is taken over the whole x range of the box: clipping it to the ball per row (warm_query) saved fewer records than
  // its sqrt, two floors and eleven registers cost here, where the records are spread over the wave anyway.
  int ty = lo_c[1], tz = lo_c[2];
  unsigned mask = 0;
  auto make_mask = [&]() {
    bool vy[4], vz[4];
    float sy[4], sz[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      vy[j] = ty + j <= hi_c[1];
      sy[j] = wide ? 0.f : slab2(1, ty + j);
      vz[j] = tz + j <= hi_c[2];
      sz[j] = (DIM == 3 && !wide) ? slab2(2, tz + j) : 0.f;
    }
    mask = 0;
#pragma unroll
    for (int jz = 0; jz < (DIM == 3 ? 4 : 1); ++jz)
#pragma unroll
      for (int jy = 0; jy < 4; ++jy)
        if (vy[jy] && vz[jz] && !(sy[jy] + sz[jz] > bf)) mask |= 1u << (jz * 4 + jy);
  };
  if (alive) make_mask();
#ifdef ICP_COOP_PROFILE
  unsigned rows_total = 0;
#endif
  for (;;) {
    // ---- owner: the next rows of its box ----
    int nr = 0;
    S.rows_d[lane][0] = 0u;  // (a lane without rows reads the bounds of cell 0 below: a cached address, no branch)
    S.rows_o[lane][0] = 0u;
    while (alive && nr < kCoopRows) {
      if (mask == 0) {  // the next tile of the box: along y, then z
        ty += 4;
        if (ty > hi_c[1]) {
          ty = lo_c[1];
          tz += 4;
        }
        if (tz > hi_c[2]) {
          alive = false;
          break;
        }
        make_mask();
        continue;
      }
      const int r = __ffs((int)mask) - 1;
      mask &= mask - 1u;
      const int jy = r & 3, jz = r >> 2;
      const uint32_t rb = ((uint32_t)(tz + jz) * g.n[1] + (uint32_t)(ty + jy)) * g.n[0];
      S.rows_d[lane][nr] = rb + lo_c[0];  // (the tables of the round are written after these have been read back)
      S.rows_o[lane][nr] = rb + hi_c[0] + 1;
      ++nr;
    }
    if (__ballot(nr > 0) == 0ull) break;
#ifdef ICP_COOP_PROFILE
    rows_total += (unsigned)nr;
#endif
    COOP_STAMP(1);
    COOP_COUNT(8, 1);
    uint32_t sb[kCoopRows], se[kCoopRows];
#pragma unroll
    for (int r = 0; r < kCoopRows; ++r) {
      const int rr = r < nr ? r : 0;  // unused slots repeat row 0 (a cached address)
      sb[r] = start[S.rows_d[lane][rr]];
      se[r] = start[S.rows_o[lane][rr]];
    }
    uint32_t rd[kCoopRows], ro[kCoopRows], Q = 0;
#pragma unroll
    for (int r = 0; r < kCoopRows; ++r) {  // runs -> quads; an empty run has no quads
      // (round 6) quads START at the run's first record: only the last quad of a run reads past it (up to three records of
      // the cells behind, or the sentinels), where quads aligned to four records read up to three on either side -- a
      // quarter fewer quads for runs of three to six records (46.3 -> 43.1 us per warm search at 1M:
      // profiles/r06_search_unaligned_quads_ab.txt)
      const uint32_t nq = ((nr > r) & (se[r] > sb[r])) ? ((se[r] - sb[r] + 3) >> 2) : 0u;  // (no short circuit: every bound is loaded up front)
      ro[r] = Q;
      rd[r] = sb[r] - 4u * Q;  // (first record of flat quad j: 4 j + this, modulo 2^32)
      Q += nq;
    }
    COOP_STAMP(2);
    // ---- the flat list of this round: an exclusive prefix sum of the owners' quads ----
    // the wave's inclusive scan without the LDS pipe: DPP row shifts inside rows of 16, then the two row broadcasts
    uint32_t incl = Q;
#define ICP_SCAN_DPP(v, ctrl, rows) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rows, 0xf, true)
    incl += ICP_SCAN_DPP(incl, 0x111, 0xf);  // row_shr:1 (zero fill at the row's start)
    incl += ICP_SCAN_DPP(incl, 0x112, 0xf);  // row_shr:2
    incl += ICP_SCAN_DPP(incl, 0x114, 0xf);  // row_shr:4
    incl += ICP_SCAN_DPP(incl, 0x118, 0xf);  // row_shr:8  -> inclusive within each row of 16
    incl += ICP_SCAN_DPP(incl, 0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    incl += ICP_SCAN_DPP(incl, 0x143, 0xc);  // row_bcast:31 into rows 2 and 3
#undef ICP_SCAN_DPP
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    S.pref[lane] = incl - Q;
#pragma unroll
    for (int r = 0; r < kCoopRows; ++r) {
      S.rows_d[lane][r] = rd[r];
      S.rows_o[lane][r] = ro[r];
    }
    __syncthreads();
    COOP_STAMP(3);
    // ---- worker: quads lane, lane + 64, ... of the list ----
    // batches of kCoopItems quads per lane while more than 64 quads are left, ONE quad per lane for a rest of up to 64
    // (round 6: half of the batches are such rests, and an item without quads costs what a full one does -- 43.1 -> 41.3 us
    // per warm search; skipping the empty item by a branch INSIDE the batch was slower, the items' lookups no longer
    // interleave: profiles/r06_search_unaligned_quads_ab.txt)
    auto batch = [&](auto ni_, const uint32_t w0) {
      constexpr int NI = decltype(ni_)::value;
      COOP_COUNT(9, 1);
      GridPoint t[4 * NI];
      unsigned own[NI];
      bool has[NI];
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const uint32_t w = w0 + 64u * it + lane;
        has[it] = w < total;
        const uint32_t wc = has[it] ? w : total - 1;
        unsigned L = 0;  // the last owner whose prefix is <= wc (owners without quads share their successor's prefix)
#pragma unroll
        for (unsigned st = 32; st >= 1; st >>= 1)
          if (S.pref[L + st] <= wc) L += st;
        own[it] = L;
        const uint32_t j = wc - S.pref[L];
        uint32_t dq = S.rows_d[L][0];
#pragma unroll
        for (int r = 1; r < kCoopRows; ++r)
          if (j >= S.rows_o[L][r]) dq = S.rows_d[L][r];
        const uint4 *line = reinterpret_cast<const uint4 *>(pts) + (size_t)(uint32_t)(4u * j + dq);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint4 wv = line[u];
          t[4 * it + u].x = __uint_as_float(wv.x);
          t[4 * it + u].y = __uint_as_float(wv.y);
          t[4 * it + u].z = __uint_as_float(wv.z);
          t[4 * it + u].idx = wv.w;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const float4 oq = S.oq[own[it]];
        const uint32_t obi = S.obi[own[it]];
        bool pass[4];
        unsigned long long bal[4];
        unsigned fresh = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float fx = oq.x - t[4 * it + u].x, fy = oq.y - t[4 * it + u].y;
          float s2 = __builtin_fmaf(fy, fy, fx * fx);
          if (DIM == 3) {
            const float fz = oq.z - t[4 * it + u].z;
            s2 = __builtin_fmaf(fz, fz, s2);
          }
          pass[u] = has[it] & !(s2 > oq.w) & (t[4 * it + u].idx != obi);  // (no short circuit: three masks and-ed)
          bal[u] = __ballot(pass[u]);
          fresh += (unsigned)__popcll(bal[u]);
        }
        if (fresh == 0) continue;
        if (cnt + fresh > (unsigned)kCoopCandCap) flush();  // (the owners' thresholds may have shrunk: the screen above is still valid, merely wider)
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (pass[u]) S.cand[cnt + (unsigned)__popcll(bal[u] & below)] = make_uint2(own[it], t[4 * it + u].idx);
          cnt += (unsigned)__popcll(bal[u]);
        }
      }
    };
    {
      uint32_t w0 = 0;
      for (; w0 + 64u < total; w0 += 64u * kCoopItems) batch(std::integral_constant<int, kCoopItems>{}, w0);
      if (w0 < total) batch(std::integral_constant<int, 1>{}, w0);
    }
    __syncthreads();  // the list and the row tables are rewritten by the next round
    COOP_STAMP(4);
    if (cnt >= (unsigned)kCoopFlushAt) {
      flush();
      COOP_STAMP(5);
    }
  }
  COOP_STAMP(1);
  if (cnt > 0) flush();
  COOP_STAMP(5);
#ifdef ICP_COOP_PROFILE
  auto coop_report = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    cprof[6] += (unsigned long long)(wall_clock64() - ct_last);
    if (in_range) atomicAdd(&g_coop_rows[rows_total < 31u ? rows_total : 31u], 1ull);
    if (lane == 0) {
      for (int j = 0; j < 12; ++j) atomicAdd(&g_coop_prof[blockIdx.x & 63][j], cprof[j]);
      if (blockIdx.x < 16384u) {
        g_coop_span[blockIdx.x][0] = ct_first;
        g_coop_span[blockIdx.x][1] = wall_clock64();
      }
    }
  };
#endif

  auto emit = [&]() {
    if (!in_range) return;
    if (no_match) {
      if (SEEDED) prev[k] = pm;
      if (idx) idx[i] = 0;
      if (a) a[i] = make_double2(q[0], q[1]);
      if (b) b[i] = make_double2(dst[0], dst[1]);
      return;
    }
    if (bi == 0xffffffffu) {  // no finite distance at all: index 0, as a scan from 0 would
      bi = 0;
      bx = dst[0];
      by = dst[1];
      bz = DIM == 3 ? dst[2] : 0.;
    }
    if (SEEDED || bi != pm.idx) {
      PrevMatch out;
      out.x = bx;
      out.y = by;
      out.z = bz;
      out.idx = bi;
      out.pad = 0;
      prev[k] = out;
    }
    if (idx) idx[i] = bi;
    if (a) a[i] = make_double2(q[0], q[1]);
    if (b) b[i] = make_double2(bx, by);
  };
  emit();
#ifdef ICP_COOP_PROFILE
  coop_report();
#endif
}

#ifdef ICP_COOP_WAVES
#define ICP_COOP_ATTR __attribute__((amdgpu_waves_per_eu(ICP_COOP_WAVES, ICP_COOP_WAVES)))
#else
#define ICP_COOP_ATTR
#endif
template <int DIM>
__global__ __launch_bounds__(kGridThreads) ICP_COOP_ATTR void k_nn_grid_warm_coop(const double *__restrict__ src,
                                                                    const uint32_t *__restrict__ perm, unsigned n, Pose T,
                                                                    GridParams g, const uint32_t *__restrict__ start,
                                                                    const GridPoint *__restrict__ pts,
                                                                    const double *__restrict__ dst, uint32_t *__restrict__ idx,
                                                                    double2 *__restrict__ a, double2 *__restrict__ b,
                                                                    PrevMatch *prev, unsigned xcd_chunk,
                                                                    const AheadPose *__restrict__ ahead) {
  __shared__ CoopLds S;
  if (ahead) {  // a search enqueued before the host knew its pose (launch_nn_grid_ahead): the launch in front of it left it here
    if (!ahead->valid) return;
    T = ahead->T;
  }
  const unsigned k = xcd_wave(blockIdx.x, gridDim.x, xcd_chunk) * kGridThreads + threadIdx.x;
  warm_wave<DIM, false>(k, n, src, perm, T, g, start, pts, dst, idx, a, b, prev, PrevMatch{0., 0., 0., 0xffffffffu, 0u}, S);
}

// ------------------------------------------------------ certified matches ----
// From one outer iteration to the next the pose moves a query by a fraction of the point spacing, and nine matches
// in ten stay what they were (profiles/r03_nn_stability.txt).  A walk that re-derives them costs as much as one that
// finds a new match; a CERTIFICATE does not: if every target other than the previous match p was farther than c
// from the query when the certificate was made, and the query has moved by at most e since, then every other
// target is still farther than c - e -- so |q' - p| < c - e proves p is the unique nearest neighbour of q', in
// exact arithmetic and therefore (the margins below are ~1e-6 relative, the contract's f64 distances round at
// 1e-16) in the contract's (d^2, index) order too.  No neighbour is visited: the slot's record holds p's
// coordinates.  c comes out of the last walk for free (warm_query<CERT>); e is bounded per query by
//     |T' s - T s| <= ||R' - R||_F |s_xy| + |t' - t|,
// summed by the host over the searches of the snapshot in launch order (QuerySort::decay_r / decay_t: searches of
// one snapshot run on one stream), so the record stores c + decay(at creation) and a check subtracts decay(now):
// nothing is written for a query that passes.  Queries that fail go to work lists (one reservation per workgroup,
// sixteen lists: a single counter serialises at ~90 reservations per microsecond), which k_nn_walk_lists walks
// with dense waves.  Results cannot depend on any of it: a certificate only ever skips a walk whose result it proves.
constexpr int kCertLists = 16;
constexpr int kCertCtrStride = 32;  // words between the lists' counters (a 128-byte line each)
constexpr int kCertThreads = 256;   // queries per reservation

template <int DIM>
__global__ __launch_bounds__(kCertThreads) void k_nn_cert(const double *__restrict__ src,
                                                          const uint32_t *__restrict__ perm, unsigned n, Pose T,
                                                          const double *__restrict__ dst, uint32_t *__restrict__ idx,
                                                          double2 *__restrict__ a, double2 *__restrict__ b,
                                                          const PrevMatch *__restrict__ prev, CertDecay cd,
                                                          uint32_t *__restrict__ lists, unsigned list_cap,
                                                          unsigned *ctr, unsigned *ctr_next) {
  __shared__ unsigned s_cnt[kCertThreads / 64], s_base;
  const unsigned k = blockIdx.x * kCertThreads + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && threadIdx.x < kCertLists) ctr_next[threadIdx.x * kCertCtrStride] = 0;  // (the next search's counters)
  bool fail = false;
  if (k < n) {
    const unsigned i = perm ? perm[k] : k;
    double q[3];
    q[0] = src[(size_t)k * DIM + 0];
    q[1] = src[(size_t)k * DIM + 1];
    q[2] = DIM == 3 ? src[(size_t)k * DIM + 2] : 0.;
    const PrevMatch pm = prev[k];
    // an upper bound of |s_xy|
    const float s_norm = __builtin_amdgcn_sqrtf((float)(q[0] * q[0] + q[1] * q[1]) * 1.0000003f) * 1.0000003f;
    {  // Transform::transform, src/transform.rs:22-24
      const double nx = (T.r00 * q[0] + T.r01 * q[1]) + T.tx;
      const double ny = (T.r10 * q[0] + T.r11 * q[1]) + T.ty;
      q[0] = nx;
      q[1] = ny;
    }
    if (pm.idx == 0xffffffffu) {  // no finite distance was ever found (NaN query): index 0, as a scan from 0 would
      if (idx) idx[i] = 0;
      if (a) a[i] = make_double2(q[0], q[1]);
      if (b) b[i] = make_double2(dst[0], dst[1]);
    } else {
      const double ddx = q[0] - pm.x, ddy = q[1] - pm.y;
      double best = ddx * ddx + ddy * ddy;
      if (DIM == 3) {
        const double ddz = q[2] - pm.z;
        best = best + ddz * ddz;
      }
      const float bf = fmaxf((float)best * 1.0000003f, 1e-37f);   // >= best
      const float rs = __builtin_amdgcn_sqrtf(bf) * 1.0000003f;   // >= sqrt(best)
      const float moved = (cd.r_hi * s_norm + cd.t_hi) * 1.000001f;  // >= the decay now
      const float c_abs = __uint_as_float(pm.pad);
      // (NaN or infinite anything: the comparison is false and the query is searched)
      const bool pass = pm.pad != 0u && (rs + moved) * 1.000001f < c_abs;
      if (pass) {
        if (idx) idx[i] = pm.idx;
        if (a) a[i] = make_double2(q[0], q[1]);
        if (b) b[i] = make_double2(pm.x, pm.y);
      } else {
        fail = true;
      }
    }
  }
  const unsigned long long mask = __ballot(fail);
  if (lane == 0) s_cnt[wave] = (unsigned)__popcll(mask);
  __syncthreads();
  const int list = blockIdx.x % kCertLists;
  if (threadIdx.x == 0) {
    unsigned tot = 0;
#pragma unroll
    for (int w = 0; w < kCertThreads / 64; ++w) tot += s_cnt[w];
    s_base = tot ? atomicAdd(&ctr[list * kCertCtrStride], tot) : 0u;
  }
  __syncthreads();
  if (fail) {
    unsigned pos = s_base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
#pragma unroll
    for (int w = 0; w < kCertThreads / 64; ++w) pos += w < wave ? s_cnt[w] : 0u;
    if (pos < list_cap) lists[(size_t)list * list_cap + pos] = k;  // (list_cap covers every query of the list's workgroups)
  }
}

// the walk proper (with a fresh certificate) for the listed queries: blockIdx.y = list
template <int DIM>
__global__ __launch_bounds__(kGridThreads) ICP_WARM_ATTR void k_nn_walk_lists(
    const double *__restrict__ src, const uint32_t *__restrict__ perm, Pose T, GridParams g,
    const uint32_t *__restrict__ start, const GridPoint *__restrict__ pts, const double *__restrict__ dst,
    uint32_t *__restrict__ idx, double2 *__restrict__ a, double2 *__restrict__ b, PrevMatch *prev, CertDecay cd,
    const uint32_t *__restrict__ lists, unsigned list_cap, const unsigned *__restrict__ ctr) {
  const unsigned cnt = min(ctr[blockIdx.y * kCertCtrStride], list_cap);
  const unsigned j = blockIdx.x * kGridThreads + threadIdx.x;
  if (j >= cnt) return;
  const unsigned k = lists[(size_t)blockIdx.y * list_cap + j];
  warm_query<DIM, true>(k, src, perm, T, g, start, pts, dst, idx, a, b, prev, cd);
}

// ---------------------------------------------------------------- seeds ----------
// The FIRST search of a snapshot has no previous matches.  The general kernel above then starts every
// query with an infinite radius and sweeps its whole 3 x 3 x 3 block before it can prune anything
// (242 us at 1M x 1M against 86 us for a warm search).  Instead: this kernel hands every query SOME
// nearby target -- the best-screened record of its own row segment, or of the smallest block of cells
// around it that holds any record -- as if it were its previous match, and the warm kernel does the
// search proper from that radius.  Nothing here needs to be exact or even good: the warm kernel's
// result does not depend on where it starts (a poor seed only costs it time).

template <int DIM>
__device__ __forceinline__ PrevMatch seed_match(const unsigned k, const double *__restrict__ src, const Pose &T,
                                                 const GridParams &g, const uint32_t *__restrict__ start,
                                                 const GridPoint *__restrict__ pts, const double *__restrict__ dst) {
  double q[3];
  q[0] = src[(size_t)k * DIM + 0];
  q[1] = src[(size_t)k * DIM + 1];
  q[2] = DIM == 3 ? src[(size_t)k * DIM + 2] : 0.;
  {  // Transform::transform, src/transform.rs:22-24
    const double nx = (T.r00 * q[0] + T.r01 * q[1]) + T.tx;
    const double ny = (T.r10 * q[0] + T.r11 * q[1]) + T.ty;
    q[0] = nx;
    q[1] = ny;
  }
  PrevMatch out;
  out.x = out.y = out.z = 0.;
  out.idx = 0xffffffffu;  // (the warm kernel: "no finite distance", index 0)
  out.pad = 0;
  const bool finite = fabs(q[0]) <= 1.7976931348623157e308 && fabs(q[1]) <= 1.7976931348623157e308 &&
                      (DIM < 3 || fabs(q[2]) <= 1.7976931348623157e308);
  if (finite) {
    float qf[3] = {0.f, 0.f, 0.f};
    int c[3] = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      qf[d] = (float)(q[d] - g.lo[d]);
      c[d] = (int)fminf(fmaxf(__builtin_floorf(qf[d] * (float)g.inv_h[d]), 0.f), (float)(g.n[d] - 1));
    }
    // (round 6, profiles/r06_cold_search_ab.txt: a quarter of a cubic cell either way and eight records in flight --
    // 113.4 -> 103.4 us for the first search of a 1M-point call; half a cell / four records was round 4's)
    constexpr int kSeedDiv = 4;
    constexpr uint32_t kSeedBatch = 8;
    const int hx = g.fx > kSeedDiv ? g.fx / kSeedDiv : 1;
    float best = __builtin_huge_valf();
    uint32_t bi = 0xffffffffu, any = 0xffffffffu;  // any: a finite target whose screened distance overflowed f32
    // blocks of cells growing around the query's own until one holds a record (the first: its own row,
    // a quarter of a cubic cell either way along x)
    for (int r = 0;; ++r) {
      const int wx = r * g.fx + hx;
      const int x0 = max(c[0] - wx, 0), x1 = min(c[0] + wx, g.n[0] - 1);
      const int y0 = max(c[1] - r, 0), y1 = min(c[1] + r, g.n[1] - 1);
      const int z0 = DIM == 3 ? max(c[2] - r, 0) : 0, z1 = DIM == 3 ? min(c[2] + r, g.n[2] - 1) : 0;
      for (int iz = z0; iz <= z1; ++iz)
        for (int iy = y0; iy <= y1; ++iy) {
          const uint32_t rb = ((uint32_t)iz * g.n[1] + iy) * g.n[0];
          const uint32_t s0 = start[rb + x0], e0 = start[rb + x1 + 1];
          // several records in flight (a run is a handful of records; one by one each load waited out the one before:
          // 113 us per 1M queries); a slot past the run repeats its last record, which changes nothing
          for (uint32_t j0 = s0; j0 < e0; j0 += kSeedBatch) {
            uint4 wv[kSeedBatch];
#pragma unroll
            for (uint32_t u = 0; u < kSeedBatch; ++u) wv[u] = reinterpret_cast<const uint4 *>(pts)[min(j0 + u, e0 - 1)];
#pragma unroll
            for (uint32_t u = 0; u < kSeedBatch; ++u) {
              const uint4 w = wv[u];
              const float fx = qf[0] - __uint_as_float(w.x), fy = qf[1] - __uint_as_float(w.y);
              const float fz = DIM == 3 ? qf[2] - __uint_as_float(w.z) : 0.f;
              const float s2 = fx * fx + fy * fy + fz * fz;
              if (s2 < best) {  // (a target with a NaN or infinite coordinate never compares less: it cannot be a seed,
                best = s2;      // and a seed at a NaN distance would never be displaced by the warm kernel)
                bi = w.w;
              } else if (s2 == __builtin_huge_valf() && any == 0xffffffffu && fabsf(__uint_as_float(w.x)) < __builtin_huge_valf() &&
                         fabsf(__uint_as_float(w.y)) < __builtin_huge_valf() && fabsf(__uint_as_float(w.z)) < __builtin_huge_valf()) {
                any = w.w;  // a finite query beyond ~1.8e19 of every target: the f32 screen overflows, the f64 distance does not
              }
            }
          }
        }
      if (bi != 0xffffffffu || any != 0xffffffffu) break;
      if (x0 == 0 && x1 == g.n[0] - 1 && y0 == 0 && y1 == g.n[1] - 1 && z0 == 0 && z1 == (DIM == 3 ? g.n[2] - 1 : 0)) break;
    }
    //  a finite query so far away that every screened distance overflowed still gets a real target: the
    // warm kernel then finds no f32 geometry for the lane and walks the grid unpruned, with the exact f64 distances
    if (bi == 0xffffffffu) bi = any;
    if (bi != 0xffffffffu) {
      out.x = dst[(size_t)bi * DIM + 0];
      out.y = dst[(size_t)bi * DIM + 1];
      out.z = DIM == 3 ? dst[(size_t)bi * DIM + 2] : 0.;
      out.idx = bi;
    }
  }
  return out;
}

template <int DIM>
__global__ __launch_bounds__(kGridThreads) void k_nn_grid_seed(const double *__restrict__ src, unsigned n, Pose T,
                                                               GridParams g, const uint32_t *__restrict__ start,
                                                               const GridPoint *__restrict__ pts,
                                                               const double *__restrict__ dst,
                                                               PrevMatch *__restrict__ prev) {
  const unsigned k = xcd_wave(blockIdx.x, gridDim.x, kXcdChunk) * kGridThreads + threadIdx.x;
  if (k >= n) return;
  prev[k] = seed_match<DIM>(k, src, T, g, start, pts, dst);
}

// The first search of a snapshot in ONE launch (round 4): the seed goes from registers straight into the warm walk --
// no 32-byte record written and read back per query, no second pass over the source cloud, one launch less per call.
template <int DIM>
__global__ __launch_bounds__(kGridThreads) void k_nn_grid_seeded(const double *__restrict__ src,
                                                                 const uint32_t *__restrict__ perm, unsigned n, Pose T,
                                                                 GridParams g, const uint32_t *__restrict__ start,
                                                                 const GridPoint *__restrict__ pts,
                                                                 const double *__restrict__ dst, uint32_t *__restrict__ idx,
                                                                 double2 *__restrict__ a, double2 *__restrict__ b,
                                                                 PrevMatch *prev) {
  __shared__ CoopLds S;
  const unsigned k = xcd_wave(blockIdx.x, gridDim.x, kXcdChunk) * kGridThreads + threadIdx.x;
  // (the lanes past the end repeat the last query: the walk is the whole wave's)
  const PrevMatch pm = seed_match<DIM>(k < n ? k : n - 1, src, T, g, start, pts, dst);
  warm_wave<DIM, true>(k, n, src, perm, T, g, start, pts, dst, idx, a, b, prev, pm, S);
}

// ------------------------------------------------ query locality (optional) -------
// Sort the source cloud by the target-grid cell of T*src, stably (ties keep the original order: the
// slot order is a pure function of the inputs, qsort.hip).  The sorted copy keeps the ORIGINAL
// coordinates (the search kernel applies the current pose with the same arithmetic as always) plus
// the permutation.  A wave's 64 queries then walk the same few cells; and because the order is
// deterministic, icp_estimate_device lets everything downstream of the search live in it
// (QuerySort::slot_order): the search stores its pairs with full-line writes instead of scattering
// them back through `perm`, and the Gauss-Newton evaluations fold them as they lie.
// `blk`: sort key = the cell index with the ROWS grouped in blocks of 2^blk x 2^blk (y, z) and the rows of a block
// interleaved -- ((block row, x cell), row in block) -- instead of row after row: 64 consecutive queries then
// come from a stretch of a 2^blk x 2^blk bundle of rows, a fraction as long as the stretch of ONE row that holds
// 64 queries, and the union of their search boxes is that much more compact (an experiment knob since the LDS-tile
// search of round 3 left the tree: HISTORY.md).  Any key is exact;
// this one only shapes the waves.
__global__ void k_query_cell(const double *__restrict__ src, unsigned n, int dim, Pose T, GridParams g, int blk,
                             uint32_t *__restrict__ cell_of, int xshift) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double q[3] = {src[(size_t)i * dim], src[(size_t)i * dim + 1], dim == 3 ? src[(size_t)i * dim + 2] : 0.};
  const double nx = (T.r00 * q[0] + T.r01 * q[1]) + T.tx;
  const double ny = (T.r10 * q[0] + T.r11 * q[1]) + T.ty;
  q[0] = nx;
  q[1] = ny;
  int c[3] = {0, 0, 0};
  for (int d = 0; d < dim; ++d) c[d] = cell_coord(q[d], g.lo[d], g.inv_h[d], g.n[d]);
  c[0] >>= xshift;  // (runs of 2^xshift cells along x share a key: fewer key bits, one sort pass less)
  const uint32_t nxk = ((uint32_t)g.n[0] + (1u << xshift) - 1u) >> xshift;
  const uint32_t nyb = ((uint32_t)g.n[1] + (1u << blk) - 1) >> blk, m = (1u << blk) - 1;
  const uint32_t brow = ((uint32_t)c[2] >> blk) * nyb + ((uint32_t)c[1] >> blk);
  const uint32_t sub = (((uint32_t)c[2] & m) << blk) | ((uint32_t)c[1] & m);
  cell_of[i] = ((brow * nxk + (uint32_t)c[0]) << (2 * blk)) | sub;
}

// one 16-byte store per lane (two consecutive doubles of the sorted array, each gathered on its own): stores of
// 8 bytes at a 24-byte stride wrote 64 MB for 24 MB of payload
__global__ void k_query_gather(const double *__restrict__ src, unsigned n, int dim, const uint32_t *__restrict__ perm,
                               double *__restrict__ sorted) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // pair of doubles
  const size_t tot = (size_t)n * dim;
  const size_t e0 = 2 * c, e1 = e0 + 1;
  if (e0 >= tot) return;
  const size_t k0 = e0 / (unsigned)dim, k1 = e1 / (unsigned)dim;
  const double v0 = src[(size_t)perm[k0] * dim + (e0 - k0 * dim)];
  if (e1 < tot) {
    const double v1 = src[(size_t)perm[k1] * dim + (e1 - k1 * dim)];
    reinterpret_cast<double2 *>(sorted)[c] = make_double2(v0, v1);
  } else {
    sorted[e0] = v0;
  }
}

__global__ void k_unpermute_idx(const uint32_t *__restrict__ slot_idx, const uint32_t *__restrict__ perm, unsigned n,
                                uint32_t *__restrict__ out) {
  const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[perm[k]] = slot_idx[k];
}

hipError_t launch_unpermute_idx(icp_handle *h, const uint32_t *d_slot_idx, size_t n, uint32_t *d_out) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_unpermute_idx, dim3(((unsigned)n + 255) / 256), dim3(256), 0, h->stream, d_slot_idx,
                     (const uint32_t *)h->qsort.d_perm, (unsigned)n, d_out);
  return hipGetLastError();
}

hipError_t prepare_queries(icp_handle *h, const double *d_src, size_t n_, const Pose &T) {
  Grid &G = h->grid;
  QuerySort &Q = h->qsort;
  Q.valid = false;
  Q.fold_n = 0;  // d_perm / d_cell_of are about to be rewritten: icp_estimate_device declares a fold order AFTER this call
  if (!G.built || n_ == 0) return hipSuccess;
  const unsigned n = (unsigned)n_;
  hipError_t e;
  hipStream_t s = h->stream;
  if (n_ > Q.cap) {
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
    (void)hipFree(Q.d_cell_of);
    (void)hipFree(Q.d_perm);
    (void)hipFree(Q.d_sorted);
    (void)hipFree(Q.d_prev);
    (void)hipFree(Q.d_cert_lists);
    (void)hipFree(Q.d_cert_ctr);
    Q.last_cert_ctr = nullptr;  // (pointed into the buffer just freed)
    Q.d_prev = nullptr;
    Q.d_cert_lists = nullptr;
    Q.d_cert_ctr = nullptr;
    Q.d_cell_of = Q.d_perm = nullptr;
    Q.d_sorted = nullptr;
    Q.cap = 0;
    Q.fold_n = 0;
    if ((e = hipMalloc(&Q.d_cell_of, n_ * 4)) != hipSuccess) return e;
    if ((e = hipMalloc(&Q.d_perm, n_ * 4)) != hipSuccess) return e;
    // three doubles per point whatever this handle's dimension: the buffers outlive it in the handle
    // pool, and a 2-D owner followed by a 3-D one of the same size must not find them short
    if ((e = hipMalloc(&Q.d_sorted, n_ * 3 * sizeof(double))) != hipSuccess) return e;
    if ((e = hipMalloc(&Q.d_prev, n_ * sizeof(PrevMatch))) != hipSuccess) return e;
    // work lists of the certified search: every list can take all the queries of its workgroups
    if ((e = hipMalloc(&Q.d_cert_lists, (n_ + (size_t)(kCertLists + 1) * kCertThreads) * sizeof(uint32_t))) != hipSuccess) return e;
    if ((e = hipMalloc(&Q.d_cert_ctr, (size_t)2 * kCertLists * kCertCtrStride * sizeof(unsigned))) != hipSuccess) return e;
    if ((e = hipMemsetAsync(Q.d_cert_ctr, 0, (size_t)2 * kCertLists * kCertCtrStride * sizeof(unsigned), s)) != hipSuccess) return e;
    Q.cert_seq = 0;
    Q.cap = n_;
  }
  // Clouds that get four lanes per query (frames of a few tens of thousands of points) keep the caller's order: the
  // snapshot exists for the per-slot previous matches, and the sort (a cell pass, the sort's launches, a gather:
  // 50 us of a 1.26 ms registration of a 28k-point frame) buys such a search nothing measurable -- the targets it
  // walks fit the L2 whatever order the queries come in.  ICP_QSORT_SMALL=1 sorts them all the same.
  static const bool sort_small = exp_env("ICP_QSORT_SMALL") != nullptr;
  // (... and so does a cloud its owner declares sorted already: the slices icp_multi_estimate deals out of the sorted cloud)
  Q.identity = ((long)n <= grid_coop_max() && !sort_small) || Q.presorted;
  Q.presorted = false;
  if (Q.identity) {
    Q.have_prev = false;
    Q.have_certs = false;
    Q.have_pose = Q.have_pose_before = false;
    Q.decay_r = Q.decay_t = 0.;
    Q.src = d_src;
    Q.n = n_;
    Q.valid = true;
    return hipSuccess;
  }
  // ICP_QSORT_BLOCK: log2 of the row bundle's side (0: row after row); the key must fit 32 bits
  // (row after row serves the gather walk best: 84.1 / 86.3 / 90.4 / 94.3 us per search for 0 / 1 / 2 / 3)
  static const int blk_env = exp_env("ICP_QSORT_BLOCK") ? atoi(exp_env("ICP_QSORT_BLOCK")) : 0;
  int blk = blk_env < 0 ? 0 : (blk_env > 3 ? 3 : blk_env);
  unsigned long long keys;
  for (;; --blk) {
    const unsigned long long nyb = ((unsigned long long)G.p.n[1] + (1ull << blk) - 1) >> blk;
    const unsigned long long nzb = ((unsigned long long)G.p.n[2] + (1ull << blk) - 1) >> blk;
    keys = (nzb * nyb * (unsigned long long)G.p.n[0]) << (2 * blk);
    if (keys <= (1ull << 32) || blk == 0) break;
  }
  // (experiments: ICP_QSORT_XSHIFT = s drops the s low bits of the x index from the sort key -- runs of 2^s x-cells share a
  // key, and the benchmark's 21-bit keys sort in two radix passes instead of three with s = 5.  Measured: 42 us less per
  // call, 3-4 us MORE per search (a wave's queries spread over a longer stretch of each row): 0.1513 against 0.1492 ms
  // per step -- the finest key stays.)
  static const int xshift_env = exp_env("ICP_QSORT_XSHIFT") ? atoi(exp_env("ICP_QSORT_XSHIFT")) : 0;
  int xshift = 0;
  if (blk == 0) {
    unsigned kb = 1;
    while (kb < 32 && (1ull << kb) < keys) ++kb;
    (void)kb;
    xshift = xshift_env > 0 ? xshift_env : 0;
    while (xshift > 0 && ((unsigned)G.p.n[0] >> xshift) == 0) --xshift;
    const unsigned nxk = ((unsigned)G.p.n[0] + (1u << xshift) - 1) >> xshift;
    keys = (unsigned long long)G.p.n[2] * (unsigned long long)G.p.n[1] * (unsigned long long)nxk;
  }
  hipLaunchKernelGGL(k_query_cell, dim3((n + 255) / 256), dim3(256), 0, s, d_src, n, h->dim, T, G.p, blk, Q.d_cell_of, xshift);
  unsigned bits = 1;
  while (bits < 32 && (1ull << bits) < keys) ++bits;
  if ((e = stable_sort_cells(Q.d_cell_of, Q.d_perm, n, bits, Q.d_tmp, Q.cap_tmp, s)) != hipSuccess) return e;
  if (Q.sort_only) {  // (a rank of a sharded registration wants the fold order of the WHOLE cloud, and only its own points out of it)
    Q.sort_only = false;
    Q.src = d_src;
    Q.n = n_;
    return hipGetLastError();  // (Q.valid stays false: no snapshot)
  }
  {
    const size_t pairs = ((size_t)n * h->dim + 1) / 2;
    hipLaunchKernelGGL(k_query_gather, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, d_src, n, h->dim,
                       (const uint32_t *)Q.d_perm, Q.d_sorted);
  }
  Q.have_prev = false;  // the first search of this snapshot reads no previous matches, it only records them
  Q.have_certs = false;
  Q.have_pose = Q.have_pose_before = false;
  Q.decay_r = Q.decay_t = 0.;
  if ((e = hipGetLastError()) != hipSuccess) return e;
  Q.src = d_src;
  Q.n = n_;
  Q.valid = true;
  return hipSuccess;
}

// four lanes per query while one lane per query cannot fill the chip (ICP_NN_COOP_MAX_N: largest n that gets them)
long grid_coop_max() {
  static const long coop_max = exp_env("ICP_NN_COOP_MAX_N") ? atol(exp_env("ICP_NN_COOP_MAX_N")) : 65536;
  return coop_max;
}

// A warm search whose pose is not known to the host yet: it is read from `d_pose` on the device, where the launch in
// front of this one on the stream leaves it (k_win_finish, AheadPose).  Only the plain shared walk qualifies (a
// snapshot of this cloud with previous matches, one lane per query, f32 geometry); *launched = false otherwise and
// nothing is enqueued.  The host cannot account for how far this pose is from the previous one, so the certificates
// of the snapshot are dropped (they would have to decay by an unknown step).
hipError_t launch_nn_grid_ahead(icp_handle *h, const double *d_src, size_t n_, const AheadPose *d_pose, double *d_a,
                                double *d_b, uint32_t *d_idx, bool *launched) {
  *launched = false;
  const Grid &G = h->grid;
  QuerySort &Q = h->qsort;
  if (n_ == 0 || n_ >= 0xffffffffull || !d_pose || h->m == 0 || !G.built || !G.p.f32_ok) return hipSuccess;
  const unsigned n = (unsigned)n_;
  if (!(Q.valid && Q.src == d_src && Q.n == n_ && Q.have_prev)) return hipSuccess;
  const bool coop = (long)n <= grid_coop_max();  // four lanes per query: the general kernel, warm
#ifdef ICP_EXPERIMENTS
  if (!coop && exp_env("ICP_NN_WARM_COOP") && atoi(exp_env("ICP_NN_WARM_COOP")) == 0) return hipSuccess;
#endif
  const double *q_src = !Q.identity ? Q.d_sorted : d_src;
  const uint32_t *q_perm = (!Q.slot_order && !Q.identity) ? Q.d_perm : nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (h->profile > 0 && (h->prof_seen++ % (unsigned)h->profile) == 0) {
    if (!h->prof_free.empty()) {
      ev0 = h->prof_free.back().first;
      ev1 = h->prof_free.back().second;
      h->prof_free.pop_back();
      (void)hipEventRecord(ev0, h->stream);
    } else if (hipEventCreate(&ev0) == hipSuccess && hipEventCreate(&ev1) == hipSuccess)
      (void)hipEventRecord(ev0, h->stream);
  }
  const unsigned blocks = (unsigned)(((size_t)n * (coop ? 4 : 1) + kGridThreads - 1) / kGridThreads);
  Q.have_pose = false;
  Q.have_certs = false;
  if (coop) {
    if (h->dim == 3)
      hipLaunchKernelGGL((k_nn_grid<3, true, false, 4>), dim3(blocks), dim3(kGridThreads), 0, h->stream, q_src, q_perm, n,
                         transform_identity(), G.p, G.d_start, G.d_pts, h->d_dst, d_idx, (double2 *)d_a, (double2 *)d_b,
                         (const PrevMatch *)Q.d_prev, Q.d_prev, d_pose);
    else
      hipLaunchKernelGGL((k_nn_grid<2, true, false, 4>), dim3(blocks), dim3(kGridThreads), 0, h->stream, q_src, q_perm, n,
                         transform_identity(), G.p, G.d_start, G.d_pts, h->d_dst, d_idx, (double2 *)d_a, (double2 *)d_b,
                         (const PrevMatch *)Q.d_prev, Q.d_prev, d_pose);
  } else if (h->dim == 3)
    hipLaunchKernelGGL(k_nn_grid_warm_coop<3>, dim3(blocks), dim3(kGridThreads), 0, h->stream, q_src, q_perm, n,
                       transform_identity(), G.p, G.d_start, G.d_pts, h->d_dst, d_idx, (double2 *)d_a, (double2 *)d_b, Q.d_prev,
                       (unsigned)kXcdChunk, d_pose);
  else
    hipLaunchKernelGGL(k_nn_grid_warm_coop<2>, dim3(blocks), dim3(kGridThreads), 0, h->stream, q_src, q_perm, n,
                       transform_identity(), G.p, G.d_start, G.d_pts, h->d_dst, d_idx, (double2 *)d_a, (double2 *)d_b, Q.d_prev,
                       (unsigned)kXcdChunk, d_pose);
  const hipError_t e = hipGetLastError();
  if (ev0 && ev1) {
    (void)hipEventRecord(ev1, h->stream);
    h->prof_events.emplace_back(ev0, ev1);
  }
  *launched = e == hipSuccess;
  return e;
}

hipError_t launch_nn_grid(icp_handle *h, const double *d_src, size_t n_, const Pose *Tp, double *d_a,
                          double *d_b, uint32_t *d_idx) {
  if (n_ == 0) return hipSuccess;
  const unsigned n = (unsigned)n_;
  const bool xform = Tp != nullptr;
  const Pose T = xform ? *Tp : transform_identity();
  const Grid &G = h->grid;
  const QuerySort &Q = h->qsort;
  const bool sorted = xform && Q.valid && Q.src == d_src && Q.n == n_;
  const double *q_src = (sorted && !Q.identity) ? Q.d_sorted : d_src;
  // slot order (icp_estimate_device): outputs stay in the snapshot's order -- k-th pair = k-th sorted point
  const uint32_t *q_perm = (sorted && !Q.slot_order && !Q.identity) ? Q.d_perm : nullptr;
  PrevMatch *q_prev_out = sorted ? Q.d_prev : nullptr;
  const PrevMatch *q_prev = (sorted && Q.have_prev) ? Q.d_prev : nullptr;
  if (sorted) h->qsort.have_prev = true;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (h->profile > 0 && (h->prof_seen++ % (unsigned)h->profile) == 0) {
    if (!h->prof_free.empty()) {
      ev0 = h->prof_free.back().first;
      ev1 = h->prof_free.back().second;
      h->prof_free.pop_back();
      (void)hipEventRecord(ev0, h->stream);
    } else if (hipEventCreate(&ev0) == hipSuccess && hipEventCreate(&ev1) == hipSuccess)
      (void)hipEventRecord(ev0, h->stream);
  }
  // four lanes per query while one lane per query cannot fill the chip (8 lanes measured the same, 16
  // slower; ICP_NN_COOP_MAX_N: largest n that gets them, 0 = never)
  const bool coop = (long)n <= grid_coop_max();
  const unsigned blocks = (unsigned)(((size_t)n * (coop ? 4 : 1) + kGridThreads - 1) / kGridThreads);
  // the warm search beyond the four-lanes-per-query sizes: the f32-geometry kernel (ICP_NN_OLD_WARM: the
  // round-1 kernel, for A/B runs; both return the same indices)
  static const bool old_warm = exp_env("ICP_NN_OLD_WARM") != nullptr;
  // (84.2 us in launch order, 83.2 / 81.1 / 82.4 / 83.1 us with chunks of 4 / 16 / 64 / 256 waves: profiles/r04_search_xcd_chunk.txt)
  static const unsigned xcd_chunk = exp_env("ICP_NN_XCD_CHUNK") ? (unsigned)atoi(exp_env("ICP_NN_XCD_CHUNK")) : kXcdChunk;
  // the first search of a snapshot: seeds, then the same warm kernel (ICP_NN_OLD_COLD: the general kernel)
  static const bool old_cold = exp_env("ICP_NN_OLD_COLD") != nullptr;
  const bool seeded = sorted && !q_prev && !coop && xform && G.p.f32_ok && !old_warm && !old_cold && h->m > 0;
  // (the seeds as a launch of their own: only where the warm search is not the plain one that takes them in registers)
  auto launch_seeds = [&]() {
    if (h->dim == 3)
      hipLaunchKernelGGL(k_nn_grid_seed<3>, dim3(blocks), dim3(kGridThreads), 0, h->stream, q_src, n, T, G.p, G.d_start,
                         G.d_pts, h->d_dst, Q.d_prev);
    else
      hipLaunchKernelGGL(k_nn_grid_seed<2>, dim3(blocks), dim3(kGridThreads), 0, h->stream, q_src, n, T, G.p, G.d_start,
                         G.d_pts, h->d_dst, Q.d_prev);
  };
  if ((q_prev || seeded) && !coop && xform && G.p.f32_ok && !old_warm) {
    // certificates (k_nn_cert above): ICP_NN_NO_CERT=1 searches every query every time, as rounds 1-2 did
    static const bool no_cert = exp_env("ICP_NN_NO_CERT") != nullptr;
    // a step so long that too few certificates survive it: skip the check (in smallest cell sides; at 0.006 a third
    // of the certificates fail, at 0.2 six in seven, and the break-even is about one half)
    static const double cert_max_step = exp_env("ICP_NN_CERT_MAX_STEP") ? atof(exp_env("ICP_NN_CERT_MAX_STEP")) : 0.01;
    QuerySort &QW = h->qsort;
    double step = 0.;
    QW.have_pose_before = QW.have_pose;
    if (QW.have_pose) {  // how far this search's pose is from the previous search's, per unit of |s_xy| and flat
      const double dr = sqrt((T.r00 - QW.last_pose.r00) * (T.r00 - QW.last_pose.r00) +
                             (T.r01 - QW.last_pose.r01) * (T.r01 - QW.last_pose.r01) +
                             (T.r10 - QW.last_pose.r10) * (T.r10 - QW.last_pose.r10) +
                             (T.r11 - QW.last_pose.r11) * (T.r11 - QW.last_pose.r11));
      const double dt = sqrt((T.tx - QW.last_pose.tx) * (T.tx - QW.last_pose.tx) +
                             (T.ty - QW.last_pose.ty) * (T.ty - QW.last_pose.ty));
      QW.decay_r += dr * (1. + 1e-12);
      QW.decay_t += dt * (1. + 1e-12);
      // (|s_xy| of a source cloud that registers against these targets: about the reach of their bounding box)
      double reach = 0.;
      for (int d = 0; d < 2; ++d) {
        const double lo = fabs(G.p.lo[d]), hi = fabs(G.p.lo[d] + G.p.h[d] * G.p.n[d]);
        reach += (lo > hi ? lo : hi) * (lo > hi ? lo : hi);
      }
      step = dr * sqrt(reach) + dt;
    }
    QW.last_pose = T;
    QW.have_pose = true;
    CertDecay cd;
    cd.r_lo = (float)(QW.decay_r * (1. - 1e-7));
    cd.t_lo = (float)(QW.decay_t * (1. - 1e-7));
    cd.r_hi = (float)(QW.decay_r * (1. + 1e-7));
    cd.t_hi = (float)(QW.decay_t * (1. + 1e-7));
    const bool decay_ok = std::isfinite(QW.decay_r) && std::isfinite(QW.decay_t);
    double hmin = G.p.h[0];
    for (int d = 1; d < h->dim; ++d) hmin = G.p.h[d] < hmin ? G.p.h[d] : hmin;
    // Certificates pay once a registration has settled (converging pair: 0.05-3 % of them fail from the tenth outer
    // iteration on, 33 % at the fourth); while the pose still moves by a tenth of a cell per iteration they do not
    // (benchmark pair: 85 % fail, and a walk that leaves certificates costs 30 % more than one that does not): such
    // searches run as rounds 1-2 had them.  Certificates already in the records stay valid either way.
    const bool certs = !no_cert && decay_ok && QW.d_cert_lists != nullptr && QW.have_pose_before && step <= cert_max_step * hmin;
    const bool check = certs && q_prev && QW.have_certs;
    // the first search of a snapshot: ONE launch, the seed handed to the walk in registers (k_nn_grid_seeded)
    static const bool no_fuse = exp_env("ICP_NN_SEED_SEPARATE") != nullptr;
#ifdef ICP_EXPERIMENTS
    static const bool warm_coop = exp_env("ICP_NN_WARM_COOP") ? atoi(exp_env("ICP_NN_WARM_COOP")) != 0 : true;
#endif
    const bool fused_seed = seeded && !check && !certs && !no_fuse;
    if (seeded && !fused_seed) launch_seeds();
    if (check) {
      const unsigned cblocks = (n + kCertThreads - 1) / kCertThreads;
      const unsigned list_cap = ((cblocks + kCertLists - 1) / kCertLists) * kCertThreads;
      unsigned *ctr = QW.d_cert_ctr + (size_t)(QW.cert_seq & 1u) * kCertLists * kCertCtrStride;
      unsigned *ctr_next = QW.d_cert_ctr + (size_t)((QW.cert_seq + 1u) & 1u) * kCertLists * kCertCtrStride;
      ++QW.cert_seq;
      ++QW.cert_searches;
      QW.last_cert_ctr = ctr;
      const dim3 wgrid((list_cap + kGridThreads - 1) / kGridThreads, kCertLists);
      if (h->dim == 3) {
        hipLaunchKernelGGL(k_nn_cert<3>, dim3(cblocks), dim3(kCertThreads), 0, h->stream, q_src, q_perm, n, T, h->d_dst,
                           d_idx, (double2 *)d_a, (double2 *)d_b, (const PrevMatch *)Q.d_prev, cd, QW.d_cert_lists,
                           list_cap, ctr, ctr_next);
        hipLaunchKernelGGL(k_nn_walk_lists<3>, wgrid, dim3(kGridThreads), 0, h->stream, q_src, q_perm, T, G.p, G.d_start,
                           G.d_pts, h->d_dst, d_idx, (double2 *)d_a, (double2 *)d_b, Q.d_prev, cd,
                           (const uint32_t *)QW.d_cert_lists, list_cap, (const unsigned *)ctr);
      } else {
        hipLaunchKernelGGL(k_nn_cert<2>, dim3(cblocks), dim3(kCertThreads), 0, h->stream, q_src, q_perm, n, T, h->d_dst,
                           d_idx, (double2 *)d_a, (double2 *)d_b, (const PrevMatch *)Q.d_prev, cd, QW.d_cert_lists,
                           list_cap, ctr, ctr_next);
        hipLaunchKernelGGL(k_nn_walk_lists<2>, wgrid, dim3(kGridThreads), 0, h->stream, q_src, q_perm, T, G.p, G.d_start,
                           G.d_pts, h->d_dst, d_idx, (double2 *)d_a, (double2 *)d_b, Q.d_prev, cd,
                           (const uint32_t *)QW.d_cert_lists, list_cap, (const unsigned *)ctr);
      }
    } else if (certs) {
      QW.have_certs = true;
      if (h->dim == 3)
        hipLaunchKernelGGL((k_nn_grid_warm<3, true>), dim3(blocks), dim3(kGridThreads), 0, h->stream, q_src, q_perm, n, T,
                           G.p, G.d_start, G.d_pts, h->d_dst, d_idx, (double2 *)d_a, (double2 *)d_b, Q.d_prev, cd, xcd_chunk);
      else
        hipLaunchKernelGGL((k_nn_grid_warm<2, true>), dim3(blocks), dim3(kGridThreads), 0, h->stream, q_src, q_perm, n, T,
                           G.p, G.d_start, G.d_pts, h->d_dst, d_idx, (double2 *)d_a, (double2 *)d_b, Q.d_prev, cd, xcd_chunk);
    } else if (fused_seed) {
      if (h->dim == 3)
        hipLaunchKernelGGL(k_nn_grid_seeded<3>, dim3(blocks), dim3(kGridThreads), 0, h->stream, q_src, q_perm, n, T, G.p,
                           G.d_start, G.d_pts, h->d_dst, d_idx, (double2 *)d_a, (double2 *)d_b, Q.d_prev);
      else
        hipLaunchKernelGGL(k_nn_grid_seeded<2>, dim3(blocks), dim3(kGridThreads), 0, h->stream, q_src, q_perm, n, T, G.p,
                           G.d_start, G.d_pts, h->d_dst, d_idx, (double2 *)d_a, (double2 *)d_b, Q.d_prev);
#ifdef ICP_EXPERIMENTS
    } else if (!warm_coop) {  // ICP_NN_WARM_COOP=0: a lane per query from end to end (rounds 2-3; same indices)
      if (h->dim == 3)
        hipLaunchKernelGGL((k_nn_grid_warm<3, false>), dim3(blocks), dim3(kGridThreads), 0, h->stream, q_src, q_perm, n, T,
                           G.p, G.d_start, G.d_pts, h->d_dst, d_idx, (double2 *)d_a, (double2 *)d_b, Q.d_prev, cd, xcd_chunk);
      else
        hipLaunchKernelGGL((k_nn_grid_warm<2, false>), dim3(blocks), dim3(kGridThreads), 0, h->stream, q_src, q_perm, n, T,
                           G.p, G.d_start, G.d_pts, h->d_dst, d_idx, (double2 *)d_a, (double2 *)d_b, Q.d_prev, cd, xcd_chunk);
#endif
    } else {
      if (h->dim == 3)
        hipLaunchKernelGGL(k_nn_grid_warm_coop<3>, dim3(blocks), dim3(kGridThreads), 0, h->stream, q_src, q_perm, n, T,
                           G.p, G.d_start, G.d_pts, h->d_dst, d_idx, (double2 *)d_a, (double2 *)d_b, Q.d_prev, xcd_chunk,
                           (const AheadPose *)nullptr);
      else
        hipLaunchKernelGGL(k_nn_grid_warm_coop<2>, dim3(blocks), dim3(kGridThreads), 0, h->stream, q_src, q_perm, n, T,
                           G.p, G.d_start, G.d_pts, h->d_dst, d_idx, (double2 *)d_a, (double2 *)d_b, Q.d_prev, xcd_chunk,
                           (const AheadPose *)nullptr);
    }
    hipError_t we = hipGetLastError();
    if (ev0 && ev1) {
      (void)hipEventRecord(ev1, h->stream);
      h->prof_events.emplace_back(ev0, ev1);
    }
    return we;
  }
#define GRID(DIM, XF)                       \
  do {                                      \
    if (q_prev) {                           \
      GRID2(DIM, XF, false);                \
    } else {                                \
      GRID2(DIM, XF, true);                 \
    }                                       \
  } while (0)
#define GRID2(DIM, XF, CD)                  \
  do {                                      \
    if (coop) {                             \
      GRID3(DIM, XF, CD, 4);                \
    } else {                                \
      GRID3(DIM, XF, CD, 1);                \
    }                                       \
  } while (0)
#define GRID3(DIM, XF, CD, LN)                                                                          \
  hipLaunchKernelGGL((k_nn_grid<DIM, XF, CD, LN>), dim3(blocks), dim3(kGridThreads), 0, h->stream, q_src, q_perm, n, T, \
                     G.p, G.d_start, G.d_pts, h->d_dst, d_idx, (double2 *)d_a, (double2 *)d_b, q_prev, q_prev_out,      \
                     (const AheadPose *)nullptr)
  if (h->dim == 3) {
    if (xform) { GRID(3, true); } else { GRID(3, false); }
  } else {
    if (xform) { GRID(2, true); } else { GRID(2, false); }
  }
#undef GRID
#undef GRID2
#undef GRID3
  hipError_t e = hipGetLastError();
  if (ev0 && ev1) {
    (void)hipEventRecord(ev1, h->stream);
    h->prof_events.emplace_back(ev0, ev1);
  }
  return e;
}

#ifdef ICP_NN_STATS
extern "C" int icp_debug_nn_hist(unsigned long long out[64], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nn_hist), 64 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[64] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_nn_hist), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}

#endif
#ifdef ICP_COOP_PROFILE
extern "C" int icp_debug_coop_rows(unsigned long long out[32], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_coop_rows), 32 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_coop_rows), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
extern "C" int icp_debug_coop_spans(long long *out, int n) {  // n <= 16384 (start, end) pairs by workgroup
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_coop_span), (size_t)n * 2 * sizeof(long long)) == hipSuccess ? 0 : 1;
}
extern "C" int icp_debug_coop_profile(unsigned long long out[12], int reset) {
  unsigned long long all[64][12];
  if (hipMemcpyFromSymbol(all, HIP_SYMBOL(g_coop_prof), sizeof(all)) != hipSuccess) return 1;
  for (int j = 0; j < 12; ++j) {
    out[j] = 0;
    for (int i = 0; i < 64; ++i) out[j] += all[i][j];
  }
  if (reset) {
    memset(all, 0, sizeof(all));
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_coop_prof), all, sizeof(all)) != hipSuccess) return 1;
  }
  return 0;
}
#endif
#ifdef ICP_NN_STATS
extern "C" int icp_debug_nn_warm(unsigned long long out[8 + 64], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nn_warm), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(g_nn_warm_hist), 64 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[64] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_nn_warm), z, 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_nn_warm_hist), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}

extern "C" int icp_debug_nn_stats(unsigned long long out[8], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nn_stats), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    const unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_nn_stats), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif

}  // namespace icp
