// The one rocPRIM algorithm the library still uses: the 64-bit radix sort of the Morton keys under the nearest-neighbour BVH
// (k_store_nearest.hip: mpg_k_build_bvh -- the tree that finishes what the index bins cannot vouch for, and the whole search on
// grids without a projection).  In a translation unit of its own so that the code object every job needs for its Stores does not
// carry rocPRIM's per-architecture instantiations (profiles/r05_init_breakdown.md); mpg_init's helper thread loads this one last.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "mpg_internal.h"

__global__ void k_sort_anchor() {}

// keys_out / vals_out = (keys_in, vals_in) sorted by key, stable
int mpg_sort_pairs_u64_i32(const unsigned long long *keys_in, unsigned long long *keys_out, const int32_t *vals_in, int32_t *vals_out, int64_t n,
                           hipStream_t s) {
  if (n <= 0) return MPG_SUCCESS;
  size_t tmp_bytes = 0;
  MPG_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, 63, s));
  TmpBuf<char> tmp;
  int rc;
  if ((rc = tmp.alloc(tmp_bytes + 16, s))) return rc;
  MPG_HIP(rocprim::radix_sort_pairs((void *)tmp.p, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, 63, s));
  return MPG_SUCCESS;
}

const void *mpg_anchor_k_sort() { return (const void *)k_sort_anchor; }
