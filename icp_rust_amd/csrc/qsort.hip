// Stable sort of the source points by target-grid cell: the once-per-estimate-call plumbing behind the
// cell-sorted source snapshot (nn_grid.hip:prepare_queries).  rocPRIM's LSD radix sort is stable, so
// sorting the cell keys alone with the values 0 .. n-1 orders the points by (cell, original index):
// a pure function of the inputs, which is what lets the snapshot order double as the order in which
// the Gauss-Newton sums are folded.  A utility primitive, not a hot kernel: one
// call per 20 outer iterations; the hot kernels of this library are hand-written.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

namespace icp {

hipError_t stable_sort_cells(const uint32_t *keys_in, uint32_t *keys_out, uint32_t *perm_out, unsigned n, unsigned bits,
                             void *&tmp, size_t &cap_tmp, hipStream_t s) {
  if (n == 0) return hipSuccess;
  if (bits < 1) bits = 1;
  if (bits > 32) bits = 32;
  rocprim::counting_iterator<uint32_t> iota(0u);
  // (the default configuration merge-sorts up to 2^20 items -- ~20 launches, ~150 us for the benchmark's 10^6
  // points; Onesweep needs three passes for its 21-bit keys.  Both are stable.)
  using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 32768>;
  size_t need = 0;
  hipError_t e = rocprim::radix_sort_pairs<cfg>(nullptr, need, keys_in, keys_out, iota, perm_out, n, 0u, bits, s);
  if (e != hipSuccess) return e;
  if (need > cap_tmp || !tmp) {
    if (tmp) {
      if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;  // an earlier sort may still use it
      (void)hipFree(tmp);
      tmp = nullptr;
      cap_tmp = 0;
    }
    const size_t want = need + need / 8 + 256;
    if ((e = hipMalloc(&tmp, want)) != hipSuccess) return e;
    cap_tmp = want;
  }
  return rocprim::radix_sort_pairs<cfg>(tmp, need, keys_in, keys_out, iota, perm_out, n, 0u, bits, s);
}

}  // namespace icp
