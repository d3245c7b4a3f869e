// Warm grid search with the targets staged into LDS tiles, one wave per tile (round 3).
//
// The warm search of nn_grid.hip (k_nn_grid_warm) gives every lane a query and lets it gather its own
// cell-table bounds and its own 64-byte record quads: 96 vector-memory instructions per wave, nearly all
// of them 64-address gathers -- the kernel is bound by the texture addresser and by the lanes of a wave
// waiting for its slowest one, while the queries of a wave (cell-sorted: QuerySort) walk nearly the same
// rows.  This kernel is the north_star's "target points staged into LDS tiles, coalesced HBM reads" applied
// to the grid:
//
//   1. every lane loads its query and its previous match (coalesced), and derives the cell box of the ball
//      around the previous match exactly as k_nn_grid_warm does (same f32 geometry, same margins);
//   2. the wave forms the UNION of its lanes' boxes -- two unions, split where the cell-sorted order wraps
//      from the end of one row of cells to the start of the next -- and, per union row (iy, iz), the slice
//      of the cell table that covers the union's x-range (strided when the range is long) is loaded with
//      coalesced reads into LDS;
//   3. the rows' record runs [start[first cell], start[last cell + 1]) are copied into LDS with
//      global_load_lds (64 consecutive 16-byte records per instruction, all rows in flight together);
//   4. every lane walks ITS box in LDS: per row the clipped window of cells -> a record range from the staged
//      table -> ds_read_b128 records screened in f32 against the lane's threshold; survivors are parked (two
//      per lane) and evaluated exactly -- the contract's f64 distance from `dst`, ties to the lowest index --
//      in one gather round trip at the end.
//
// The staged table may be coarser than the cells and the windows wider than the ball: a lane can only see
// MORE records than k_nn_grid_warm's walk would show it, all of them real targets, and every record whose
// screened distance does not exceed the lane's threshold is evaluated exactly -- the result is the same
// exact minimum by (d^2, index), bit for bit (tests/test_gpu_parity.py, 1M x 1M included).
// A wave whose unions do not fit the LDS budget (rows, table entries, records), or that holds a lane without
// f32 geometry, appends its number to a worklist and leaves; k_nn_grid_warm_list (nn_grid.hip) then serves
// those waves with the per-lane gather walk.
// (an EXPERIMENT: compiled only into `make experiments`, never into the product library)
#ifdef ICP_EXPERIMENTS
#include "common.hpp"

namespace icp {

constexpr int kTileRec = 512;   // records staged per wave (8 KB)
constexpr int kTileTab = 1024;  // cell-table entries staged per wave, rows at a power-of-two pitch (4 KB)
constexpr int kTileItems = 1024; // work items (eight records of one query's window each) per wave
constexpr int kTileSpill = 64;   // survivors beyond two per query
constexpr int kTileRows = 64;   // union rows per wave: one lane per row derives its run
// split the wave where the low x cell of consecutive boxes drops by more than this (the sorted order wraps)
constexpr int kTileWrapCells = 16;

// ---- wave-wide reductions (every lane takes part: inactive queries pass the neutral element) ----------
// rows of 16 lanes: Hillis-Steele with row_shr 1, 2, 4, 8; then row 0 -> 1, 2 -> 3 (row_bcast:15), and
// lanes 31 -> rows 2, 3 (row_bcast:31): lane 63 holds the result
template <bool MAX>
__device__ __forceinline__ int wave_reduce(int v) {
#define ICP_TILE_STEP(ctrl, rmask)                                                     \
  {                                                                                    \
    const int o = __builtin_amdgcn_update_dpp(v, v, ctrl, rmask, 0xf, false);          \
    v = MAX ? max(v, o) : min(v, o);                                                   \
  }
  ICP_TILE_STEP(0x111, 0xf)
  ICP_TILE_STEP(0x112, 0xf)
  ICP_TILE_STEP(0x114, 0xf)
  ICP_TILE_STEP(0x118, 0xf)
  ICP_TILE_STEP(0x142, 0xa)
  ICP_TILE_STEP(0x143, 0xc)
#undef ICP_TILE_STEP
  return __builtin_amdgcn_readlane(v, 63);
}

// Three wave reductions at once (results in lane 63).  A DPP operand must not have been written by either of the
// two VALU instructions before it; three interleaved chains put exactly two instructions between a step and the
// next step of the same chain.  Rows of 16 lanes: row_shr 1, 2, 4, 8 (Hillis-Steele); then row 0 -> 1 and
// 2 -> 3 (row_bcast:15), rows 0-1 -> 2-3 (row_bcast:31).  Lanes without a DPP source keep their value.
#define ICP_RED3_STEP(OP, CTRL) \
  OP " %0, %0, %0 " CTRL "\n\t" OP " %1, %1, %1 " CTRL "\n\t" OP " %2, %2, %2 " CTRL "\n\t"
#define ICP_RED3(OP, x, y, z)                                                                      \
  asm volatile("s_nop 1\n\t" ICP_RED3_STEP(OP, "row_shr:1 row_mask:0xf bank_mask:0xf")             \
                   ICP_RED3_STEP(OP, "row_shr:2 row_mask:0xf bank_mask:0xf")                        \
                       ICP_RED3_STEP(OP, "row_shr:4 row_mask:0xf bank_mask:0xf")                    \
                           ICP_RED3_STEP(OP, "row_shr:8 row_mask:0xf bank_mask:0xf")                \
                               ICP_RED3_STEP(OP, "row_bcast:15 row_mask:0xa bank_mask:0xf")         \
                                   ICP_RED3_STEP(OP, "row_bcast:31 row_mask:0xc bank_mask:0xf")     \
               "s_nop 1"                                                                           \
               : "+v"(x), "+v"(y), "+v"(z))

// floor(a / d) for a < 2^16, 0 < d <= 2^12 (float quotient, corrected)
__device__ __forceinline__ unsigned small_div(unsigned a, unsigned d, float inv_d) {
  unsigned q = (unsigned)(((float)a + 0.5f) * inv_d);
  if (q * d > a) --q;
  else if ((q + 1) * d <= a) ++q;
  return q;
}

#ifdef ICP_TILE_PROFILE
// Diagnostic build only (make stats): shader-clock stamps per wave and phase -- [0] start, [1] query + previous
// match loaded / box known, [2] unions known, [3] table staged, [4] records staged, [5] walk done, [6] exact
// evaluations done, [7] end; a wave that gave up leaves [7] = 0.
__device__ unsigned long long g_tile_prof[16384][8];
#define TILE_STAMP(i) do { if (lane == 0 && w < 16384u) g_tile_prof[w][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TILE_STAMP(i) ((void)0)
#endif

template <int DIM>
__global__ __launch_bounds__(64) void k_nn_tile(const double *__restrict__ src, const uint32_t *__restrict__ perm,
                                                unsigned n, Pose T, GridParams g, const uint32_t *__restrict__ start,
                                                const GridPoint *__restrict__ pts, const double *__restrict__ dst,
                                                uint32_t *__restrict__ idx, double2 *__restrict__ a,
                                                double2 *__restrict__ b, PrevMatch *prev, uint32_t *__restrict__ flags,
                                                unsigned nwaves, int xcd_chunks) {
  __shared__ uint4 s_rec[kTileRec + 8];  // (+ eight +inf sentinels behind the staged records)
  __shared__ uint32_t s_tab[kTileTab];
  __shared__ int s_rowdelta[kTileRows];
  __shared__ float4 s_q[64];            // per query: f32 position relative to the grid origin, screen threshold
  __shared__ uint32_t s_qbi[64];        // per query: its current match (never a survivor)
  __shared__ uint16_t s_item[kTileItems];
  __shared__ unsigned s_qn[64];         // per query: survivors so far
  __shared__ uint32_t s_qc[128];        // per query: its first two survivors
  __shared__ uint2 s_spill[kTileSpill]; // survivors beyond two per query: (query, target)
  __shared__ unsigned s_cnt[2];         // [1]: spilled survivors
  const unsigned lane = threadIdx.x;
  // workgroups are dealt round-robin over the 8 XCDs: give each XCD a contiguous eighth of the sorted
  // queries, so that its L2 holds the eighth of the cell table and of the records they share (speed only)
  unsigned w = blockIdx.x;
  if (xcd_chunks > 0) w = (blockIdx.x & 7u) * (unsigned)xcd_chunks + (blockIdx.x >> 3);
  if (w >= nwaves) return;
  s_qn[lane] = 0u;
  if (lane < 2) s_cnt[lane] = 0u;
#ifdef ICP_TILE_PROFILE
  if (lane == 0 && w < 16384u) g_tile_prof[w][7] = 0;
#endif
  TILE_STAMP(0);
  const unsigned k = w * 64u + lane;
  const bool live = k < n;
  const unsigned kq = live ? k : n - 1;
  double q[3];
  q[0] = src[(size_t)kq * DIM + 0];
  q[1] = src[(size_t)kq * DIM + 1];
  q[2] = DIM == 3 ? src[(size_t)kq * DIM + 2] : 0.;
  {  // Transform::transform, src/transform.rs:22-24
    const double nx = (T.r00 * q[0] + T.r01 * q[1]) + T.tx;
    const double ny = (T.r10 * q[0] + T.r11 * q[1]) + T.ty;
    q[0] = nx;
    q[1] = ny;
  }
  const PrevMatch pm = prev[kq];
  const bool matched = live && pm.idx != 0xffffffffu;  // (no match: "NaN query", index 0 -- see the epilogue)
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
  // ---- f32 geometry relative to the grid origin: the definitions of k_nn_grid_warm (nn_grid.hip), unchanged ----
  float qf[3], amax = 0.f;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    qf[d] = d < DIM ? (float)(q[d] - g.lo[d]) : 0.f;
    amax = fmaxf(amax, fabsf(qf[d]));
  }
  const float mgf = 4e-7f * (amax + g.ext);
  const float ecf = 2.1e-7f * (amax + g.ext);
  float bf, rf, thr32;  // bf >= best; rf >= sqrt(best) + mgf; records with s32 > thr32 cannot win or tie
  auto set_radius = [&]() {
    bf = fmaxf((float)best * 1.0000003f, 1e-37f);
    const float rs = __builtin_amdgcn_sqrtf(bf) * 1.0000003f;
    rf = rs + mgf;
    thr32 = (rs + ecf) * (rs + ecf) * 1.000005f;
  };
  set_radius();
  const bool wide = !(amax + rf < 1e18f);  // (also NaN): no f32 geometry for this lane
  const float ihf[3] = {(float)g.inv_h[0], (float)g.inv_h[1], (float)g.inv_h[2]};
  auto cell_lo = [&](float v, float em, int d) -> int {
    const float t = fminf(fmaxf(__builtin_floorf(v * ihf[d] - em), 0.f), (float)(g.n[d] - 1));
    return (int)t;
  };
  auto cell_hi = [&](float v, float em, int d) -> int {
    const float t = fminf(fmaxf(__builtin_floorf(v * ihf[d] + em), 0.f), (float)(g.n[d] - 1));
    return (int)t;
  };
  // a wave with a lane whose geometry does not fit f32 goes to the gather walk as a whole (adversarial inputs only)
  bool give_up = __ballot(matched && wide) != 0ull;
  int lo_c[3] = {0, 0, 0}, hi_c[3] = {0, 0, 0};
  float em[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    em[d] = (fabsf(qf[d]) + rf) * ihf[d] * 4e-7f + 1e-3f;
    lo_c[d] = cell_lo(qf[d] - rf, em[d], d);
    hi_c[d] = cell_hi(qf[d] + rf, em[d], d);
  }

#ifdef ICP_TILE_PROFILE
  if (__ballot(bf > 0.f) == 0ull) return;  // (never: keeps the loads above ahead of the stamp)
#endif
  TILE_STAMP(1);
  // ---- the wave's two unions of boxes ----
  int split = 64;
  {
    const int vx = matched ? lo_c[0] : -1;
    const int px = __shfl_up(vx, 1);  // (lane 0 keeps its own)
    const int drop = (lane > 0 && vx >= 0 && px >= 0) ? px - vx : 0;
    int key = drop > kTileWrapCells ? ((drop << 6) | (int)lane) : 0, k1 = 0, k2 = 0;
    ICP_RED3("v_max_i32_dpp", key, k1, k2);
    key = __builtin_amdgcn_readlane(key, 63);
    if (key) split = key & 63;
  }
  const int grp = (int)lane >= split ? 1 : 0;
  // twelve reductions, three interleaved chains at a time (wave_reduce3)
  int a0 = (matched && !grp) ? lo_c[0] : 0x7fffffff, a1 = (matched && !grp) ? lo_c[1] : 0x7fffffff,
      a2 = (matched && !grp) ? lo_c[2] : 0x7fffffff;
  int b0 = (matched && !grp) ? hi_c[0] : -1, b1 = (matched && !grp) ? hi_c[1] : -1, b2 = (matched && !grp) ? hi_c[2] : -1;
  int c0 = (matched && grp) ? lo_c[0] : 0x7fffffff, c1 = (matched && grp) ? lo_c[1] : 0x7fffffff,
      c2 = (matched && grp) ? lo_c[2] : 0x7fffffff;
  int d0 = (matched && grp) ? hi_c[0] : -1, d1 = (matched && grp) ? hi_c[1] : -1, d2 = (matched && grp) ? hi_c[2] : -1;
  ICP_RED3("v_min_i32_dpp", a0, a1, a2);
  ICP_RED3("v_max_i32_dpp", b0, b1, b2);
  ICP_RED3("v_min_i32_dpp", c0, c1, c2);
  ICP_RED3("v_max_i32_dpp", d0, d1, d2);
  // (scalars per union, selected with ICP_U: register arrays with a run-time index would live in scratch memory)
  int x0A = __builtin_amdgcn_readlane(a0, 63), y0A = __builtin_amdgcn_readlane(a1, 63), z0A = __builtin_amdgcn_readlane(a2, 63);
  const int x1A = __builtin_amdgcn_readlane(b0, 63), y1A = __builtin_amdgcn_readlane(b1, 63), z1A = __builtin_amdgcn_readlane(b2, 63);
  int x0B = __builtin_amdgcn_readlane(c0, 63), y0B = __builtin_amdgcn_readlane(c1, 63), z0B = __builtin_amdgcn_readlane(c2, 63);
  const int x1B = __builtin_amdgcn_readlane(d0, 63), y1B = __builtin_amdgcn_readlane(d1, 63), z1B = __builtin_amdgcn_readlane(d2, 63);
#define ICP_U(gi, v0, v1) ((gi) ? (v1) : (v0))
  const bool anyA = x1A >= 0, anyB = x1B >= 0;
  const int nyA = anyA ? y1A - y0A + 1 : 0, nzA = anyA ? (DIM == 3 ? z1A - z0A + 1 : 1) : 0;
  const int nyB = anyB ? y1B - y0B + 1 : 0, nzB = anyB ? (DIM == 3 ? z1B - z0B + 1 : 1) : 0;
  const int nrowA = nyA * nzA, nrowB = nyB * nzB;
  if (!anyA) x0A = y0A = z0A = 0;
  if (!anyB) x0B = y0B = z0B = 0;
  if (DIM < 3) z0A = z0B = 0;
  const int R = nrowA + nrowB;
  if (R > kTileRows) give_up = true;
  // Table stride 2^sh: the smallest that fits the slices of both unions into the staged table.  A row's slice has
  // ent entries (entry j <-> cell min(x0 + (j << sh), nx), nx = the row's end) and lives at a power-of-two pitch,
  // so that an entry's row and column are a shift and a mask.
  int sh = 0, entA = 0, entB = 0, esh = 0;
  if (!give_up && R > 0) {
    const int cellsA = anyA ? x1A - x0A + 1 : 0, cellsB = anyB ? x1B - x0B + 1 : 0;
    for (;; ++sh) {
      entA = anyA ? ((cellsA + (1 << sh) - 1) >> sh) + 1 : 0;
      entB = anyB ? ((cellsB + (1 << sh) - 1) >> sh) + 1 : 0;
      const int em_ = max(entA, entB);
      esh = 32 - __builtin_clz((unsigned)(em_ - 1) | 1u);  // pitch 2^esh >= em_ (em_ >= 2)
      if ((R << esh) <= kTileTab || sh >= 14) break;
    }
    if ((R << esh) > kTileTab) give_up = true;
  }
  // (a flag per wave, not an appended list: a third of the waves appending through one counter queued for
  // ~60 us on it -- a word serves ~88 returning atomics per microsecond)
  if (give_up) {
    if (lane == 0) flags[w] = 1u;
    return;
  }
  TILE_STAMP(2);
  if (R > 0) {
    // ---- one lane per union row: the cell index of the row's first cell ----
    if ((int)lane < R) {
      const int gi = (int)lane >= nrowA ? 1 : 0;
      const unsigned rr = (unsigned)((int)lane - (gi ? nrowA : 0));
      const unsigned ny_ = (unsigned)ICP_U(gi, nyA, nyB);
      const unsigned rz = small_div(rr, ny_, 1.f / (float)ny_);
      const unsigned ry = rr - rz * ny_;
      s_rowdelta[lane] = (int)(((unsigned)(ICP_U(gi, z0A, z0B) + (int)rz) * (unsigned)g.n[1] +
                                (unsigned)(ICP_U(gi, y0A, y0B) + (int)ry)) * (unsigned)g.n[0]);
    }
    __syncthreads();
    // ---- stage the cell-table slices (four loads in flight per lane) ----
    const int tot = R << esh, emask = (1 << esh) - 1;
    for (int base = 0; base < tot; base += 256) {
      uint32_t tv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = base + u * 64 + (int)lane;
        const int r = t >> esh, j = t & emask;
        const int gi = r >= nrowA ? 1 : 0;
        tv[u] = 0;
        if (t < tot && j < ICP_U(gi, entA, entB)) {
          const unsigned cx = min((unsigned)ICP_U(gi, x0A, x0B) + ((unsigned)j << sh), (unsigned)g.n[0]);
          tv[u] = start[(unsigned)s_rowdelta[r] + cx];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = base + u * 64 + (int)lane;
        if (t < tot) s_tab[t] = tv[u];
      }
    }
    __syncthreads();
    TILE_STAMP(3);
    // ---- one lane per union row: its run of records, and where it goes in LDS ----
    uint32_t S = 0, len = 0;
    if ((int)lane < R) {
      const int e_ = (int)lane >= nrowA ? entB : entA;
      S = s_tab[(int)lane << esh];
      len = s_tab[((int)lane << esh) + e_ - 1] - S;
    }
    uint32_t inc = len;  // inclusive prefix sum over the lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)inc, off);
      if ((int)lane >= off) inc += t;
    }
    const uint32_t U = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    if (U > (uint32_t)kTileRec) {  // (uniform)
      if (lane == 0) flags[w] = 1u;
      return;
    }
    const uint32_t off_r = inc - len;
    __syncthreads();  // (s_rowdelta is reused: every lane has read its row's cell index)
    if ((int)lane < R) s_rowdelta[lane] = (int)off_r - (int)S;
    // ---- stage the records: 64 consecutive records per instruction straight into LDS, everything in flight ----
    const uint4 *pts4 = reinterpret_cast<const uint4 *>(pts);
    for (int r = 0; r < R; ++r) {
      const uint32_t Sr = (uint32_t)__builtin_amdgcn_readlane((int)S, r);
      const uint32_t lr = (uint32_t)__builtin_amdgcn_readlane((int)len, r);
      const uint32_t or_ = (uint32_t)__builtin_amdgcn_readlane((int)off_r, r);
      for (uint32_t c0_ = 0; c0_ < lr; c0_ += 64u) {
        if (c0_ + lane < lr)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pts4 + Sr + c0_ + lane),
                                           (__attribute__((address_space(3))) void *)(s_rec + or_ + c0_), 16, 0, 0);
      }
    }
    if (lane < 8)  // behind the staged records: items read eight records whatever their window's length
      s_rec[U + lane] = make_uint4(0x7f800000u, 0x7f800000u, 0x7f800000u, 0u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  s_q[lane] = make_float4(qf[0], qf[1], qf[2], matched ? thr32 : -1.f);  // (-1: nothing passes for a lane without a search)
  s_qbi[lane] = bi;
  __syncthreads();
  TILE_STAMP(4);

  // ---- work items: every (query, row of its box) window, cut into blocks of eight staged records ----
  // A lane walking its own windows would make the wave wait for its busiest lane (the windows of a wave's
  // queries differ several-fold in length).  Instead the windows become work items in LDS -- 16 bits each:
  // the query's lane and the position of eight consecutive staged records -- and the wave then screens the
  // items 64 at a time, whichever queries they belong to.  No per-row clipping to the ball here: a few
  // instructions per (lane, row) cost the whole wave, eight more records cost one lane-slot of one pass.
  // Two passes over a lane's rows -- count, then write behind an exclusive prefix sum over the lanes -- so that
  // no lane waits for an LDS atomic per row (a first version reserved its slots that way: 21 000 cycles per wave).
  const int x0 = ICP_U(grp, x0A, x0B), y0 = ICP_U(grp, y0A, y0B), z0 = ICP_U(grp, z0A, z0B);
  const int nyg = ICP_U(grp, nyA, nyB), rbase = grp ? nrowA : 0;
  const int jl = (lo_c[0] - x0) >> sh, jh = (hi_c[0] + 1 - x0 + ((1 << sh) - 1)) >> sh;
  const int nrows_mine = matched ? (hi_c[1] - lo_c[1] + 1) * (hi_c[2] - lo_c[2] + 1) : 0;
  unsigned mine = 0;
  {
    int iy = lo_c[1], iz = lo_c[2];
    for (int t_ = 0; t_ < nrows_mine; ++t_) {
      const int row = rbase + (iz - z0) * nyg + (iy - y0);
      const int len = (int)s_tab[(row << esh) + jh] - (int)s_tab[(row << esh) + jl];
      mine += (unsigned)((len + 7) >> 3);
      if (++iy > hi_c[1]) iy = lo_c[1], ++iz;
    }
  }
  unsigned inc_i = mine;  // inclusive prefix sum over the lanes
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned t_ = (unsigned)__shfl_up((int)inc_i, off);
    if ((int)lane >= off) inc_i += t_;
  }
  const unsigned nitem = (unsigned)__builtin_amdgcn_readlane((int)inc_i, 63);
  if (nitem > (unsigned)kTileItems) {  // (uniform)
    if (lane == 0) flags[w] = 1u;
    return;
  }
  {
    unsigned slot = inc_i - mine;
    int iy = lo_c[1], iz = lo_c[2];
    for (int t_ = 0; t_ < nrows_mine; ++t_) {
      const int row = rbase + (iz - z0) * nyg + (iy - y0);
      const int delta = s_rowdelta[row];
      int p = (int)s_tab[(row << esh) + jl] + delta;
      const int pe = (int)s_tab[(row << esh) + jh] + delta;
      for (; p < pe; p += 8) s_item[slot++] = (uint16_t)((lane << 10) | (unsigned)p);
      if (++iy > hi_c[1]) iy = lo_c[1], ++iz;
    }
  }
  __syncthreads();
  TILE_STAMP(5);
  // ---- screen the items: a query's f32 position, threshold and current match come from LDS ----
  for (unsigned it = lane; it < nitem; it += 64u) {
    const unsigned item = s_item[it];
    const unsigned ql = item >> 10, p = item & 1023u;
    const float4 qd = s_q[ql];
    const uint32_t qb = s_qbi[ql];
    uint4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = s_rec[p + u];  // (past a window's end: further staged targets, or +inf sentinels)
    unsigned pass = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float fx = qd.x - __uint_as_float(t[u].x), fy = qd.y - __uint_as_float(t[u].y);
      float s2 = __builtin_fmaf(fy, fy, fx * fx);
      if (DIM == 3) {
        const float fz = qd.z - __uint_as_float(t[u].z);
        s2 = __builtin_fmaf(fz, fz, s2);
      }
      pass |= (!(s2 > qd.w) && t[u].w != qb) ? (1u << u) : 0u;
    }
    if (pass) {  // rare once a registration has settled: the survivor goes to its query's two slots, else to the spill list
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (pass & (1u << u)) {
          const unsigned nq = atomicAdd(&s_qn[ql], 1u);
          if (nq < 2u) s_qc[2 * ql + nq] = t[u].w;
          else {
            const unsigned o = atomicAdd(&s_cnt[1], 1u);
            if (o < (unsigned)kTileSpill) s_spill[o] = make_uint2(ql, t[u].w);
          }
        }
    }
  }
  __syncthreads();
  const unsigned nspill = s_cnt[1];
  if (nspill > (unsigned)kTileSpill) {  // (uniform) a poor starting radius: the gather walk tightens as it goes
    if (lane == 0) flags[w] = 1u;
    return;
  }
  TILE_STAMP(6);
  // ---- exact evaluation of the survivors: the contract's f64 distance, ties to the lowest index ----
  auto consider = [&](uint32_t ti) {
    const double tx = dst[(size_t)ti * DIM + 0], ty = dst[(size_t)ti * DIM + 1];
    const double tz = DIM == 3 ? dst[(size_t)ti * DIM + 2] : 0.;
    const double dd = dist2(tx, ty, tz);
    if (dd < best || (dd == best && ti < bi)) {
      best = dd;
      bi = ti;
      bx = tx;
      by = ty;
      bz = tz;
    }
  };
  if (matched) {
    const unsigned nq = s_qn[lane];
    if (nq > 0u) consider(s_qc[2 * lane]);
    if (nq > 1u) consider(s_qc[2 * lane + 1]);
    for (unsigned o = 0; o < nspill; ++o) {
      const uint2 e = s_spill[o];
      if (e.x == lane) consider(e.y);
    }
  }
#undef ICP_U
  if (lane == 0) flags[w] = 0u;
  if (!live) return;
  const unsigned i = perm ? perm[k] : k;  // null: outputs in slot order
  if (!matched || bi == 0xffffffffu) {  // no finite distance at all: index 0, as a scan from 0 would (k_nn_grid does the same)
    if (idx) idx[i] = 0;
    if (a) a[i] = make_double2(q[0], q[1]);
    if (b) b[i] = make_double2(dst[0], dst[1]);
    if (matched) {  // (a previous match at a NaN distance and nothing else found)
      PrevMatch out;
      out.x = dst[0];
      out.y = dst[1];
      out.z = DIM == 3 ? dst[2] : 0.;
      out.idx = 0;
      out.pad = 0;
      prev[k] = out;
    }
    return;
  }
  if (bi != pm.idx) {  // a slot whose match did not change already holds this record
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
#ifdef ICP_TILE_PROFILE
  if (__ballot(best >= 0.) != 0ull) TILE_STAMP(7);
#endif
}

#ifdef ICP_TILE_PROFILE
extern "C" int icp_debug_tile_profile(unsigned long long *out /* 16384 x 8 */) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_prof), sizeof(unsigned long long) * 16384 * 8) == hipSuccess ? 0 : 1;
}
#endif

hipError_t launch_nn_warm_flagged(icp_handle *h, const double *q_src, const uint32_t *q_perm, unsigned n, const Pose &T,
                                  uint32_t *d_idx, double2 *d_a, double2 *d_b, const uint32_t *flags);

hipError_t launch_nn_tile(icp_handle *h, const double *q_src, const uint32_t *q_perm, unsigned n, const Pose &T,
                          uint32_t *d_idx, double2 *d_a, double2 *d_b) {
  const Grid &G = h->grid;
  QuerySort &Q = h->qsort;
  const unsigned nwaves = (n + 63u) / 64u;
  static const bool no_xcd = exp_env("ICP_TILE_NO_XCD") != nullptr;
  const unsigned chunks = (nwaves + 7u) / 8u;
  const unsigned blocks = no_xcd ? nwaves : chunks * 8u;
  Q.last_waves = nwaves;
  if (h->dim == 3)
    hipLaunchKernelGGL(k_nn_tile<3>, dim3(blocks), dim3(64), 0, h->stream, q_src, q_perm, n, T, G.p, G.d_start, G.d_pts,
                       h->d_dst, d_idx, d_a, d_b, Q.d_prev, Q.d_list, nwaves, no_xcd ? 0 : (int)chunks);
  else
    hipLaunchKernelGGL(k_nn_tile<2>, dim3(blocks), dim3(64), 0, h->stream, q_src, q_perm, n, T, G.p, G.d_start, G.d_pts,
                       h->d_dst, d_idx, d_a, d_b, Q.d_prev, Q.d_list, nwaves, no_xcd ? 0 : (int)chunks);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // the waves the tile kernel handed back: the per-lane gather walk; every other workgroup leaves at once
  return launch_nn_warm_flagged(h, q_src, q_perm, n, T, d_idx, d_a, d_b, Q.d_list);
}

}  // namespace icp

#endif  // ICP_EXPERIMENTS
