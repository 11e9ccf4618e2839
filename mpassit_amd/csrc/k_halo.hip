// Source-halo support for the multi-GPU path (replaces the per-Regrid source exchange hidden in ESMF's
// route handle): sorted unique source ids referenced by a handle, index compaction,
// and the pack kernel lives in k_apply.hip.
#include <cstring>


#include "mpg_internal.h"

__global__ __launch_bounds__(256) void k_mark(int64_t n, const int32_t *__restrict__ idx, int32_t *__restrict__ flag) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t c = idx[i];
    if (c >= 0) flag[c] = 1;
  }
}
__global__ __launch_bounds__(256) void k_compact(int64_t n, const int32_t *__restrict__ flag, const int32_t *__restrict__ pos,
                                                 int32_t *__restrict__ ids) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (flag[i]) ids[pos[i]] = (int32_t)i;
}
__global__ __launch_bounds__(256) void k_remap(int64_t n, int32_t *__restrict__ idx, const int32_t *__restrict__ pos) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t c = idx[i];
    if (c >= 0) idx[i] = pos[c];
  }
}

int mpg_k_unique_sources(mpg_handle_s *h, std::vector<int32_t> &ids, bool remap, hipStream_t s) {
  int rc;
  int64_t n = h->n_src;
  int32_t *ip = h->kind == MPG_KIND_CSR ? h->col.p : h->idx.p;
  int64_t ni = h->kind == MPG_KIND_CSR ? h->nnz : (int64_t)h->nnz_per_row * h->n_dst;
  TmpBuf<int32_t> flag, pos, out;
  if ((rc = flag.alloc((size_t)n + 1)) || (rc = pos.alloc((size_t)n + 1))) return rc;
  MPG_HIP(hipMemsetAsync(flag.p, 0, sizeof(int32_t) * (n + 1), s));
  int gb = (int)((ni + 255) / 256);
  if (gb > 8192) gb = 8192;
  if (gb < 1) gb = 1;
  k_mark<<<gb, 256, 0, s>>>(ni, ip, flag.p);
  if ((rc = mpg_scan_excl_i32(flag.p, pos.p, n + 1, s))) return rc;
  int32_t nu = 0;
  MPG_HIP(hipMemcpyAsync(&nu, pos.p + n, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  ids.resize((size_t)nu);
  if ((rc = out.alloc((size_t)nu + 1))) return rc;
  int gn = (int)((n + 255) / 256);
  if (gn > 8192) gn = 8192;
  k_compact<<<gn, 256, 0, s>>>(n, flag.p, pos.p, out.p);
  if (nu) MPG_HIP(hipMemcpyAsync(ids.data(), out.p, sizeof(int32_t) * nu, hipMemcpyDeviceToHost, s));
  if (remap) {
    h->src_range_valid = false;   // the indices change
    k_remap<<<gb, 256, 0, s>>>(ni, ip, pos.p);
    h->n_src = nu;
    h->localized = true;
  }
  MPG_HIP(hipGetLastError());
  MPG_HIP(hipStreamSynchronize(s));
  flag.free(); pos.free(); out.free();
  return MPG_SUCCESS;
}


__global__ __launch_bounds__(256) void k_rebase(int64_t n, int32_t *__restrict__ idx, int32_t base, int32_t nlocal,
                                                int32_t *__restrict__ bad) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t c = idx[i];
    if (c < 0) continue;
    c -= base;
    if (c < 0 || c >= nlocal) atomicAdd(bad, 1);
    idx[i] = c;
  }
}

// [first, end) of the source ids a handle references (first == end: none), in its current index space
__global__ __launch_bounds__(256) void k_src_range(int64_t n, const int32_t *__restrict__ idx, int32_t *__restrict__ lohi) {
  __shared__ int32_t slo[4], shi[4];
  int32_t lo = 0x7fffffff, hi = -1;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t c = idx[i];
    if (c >= 0) {
      lo = min(lo, c);
      hi = max(hi, c);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, __shfl_down(lo, o));
    hi = max(hi, __shfl_down(hi, o));
  }
  if ((threadIdx.x & 63) == 0) {
    slo[threadIdx.x >> 6] = lo;
    shi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // one pair of atomics per workgroup, a few hundred workgroups: nothing queues on the two addresses
    lo = min(min(slo[0], slo[1]), min(slo[2], slo[3]));
    hi = max(max(shi[0], shi[1]), max(shi[2], shi[3]));
    if (hi >= 0) {
      atomicMin(lohi, lo);
      atomicMax(lohi + 1, hi);
    }
  }
}

int mpg_k_source_range(mpg_handle_s *h, int64_t *first, int64_t *end, hipStream_t s) {
  int rc;
  if (h->src_range_valid) {   // a property of the handle's indices: computed once (every host-array Regrid asks, mpg_hostpipe.hip)
    *first = h->src_range_first;
    *end = h->src_range_end;
    return MPG_SUCCESS;
  }
  int32_t *ip = h->kind == MPG_KIND_CSR ? h->col.p : h->idx.p;
  int64_t ni = h->kind == MPG_KIND_CSR ? h->nnz : (int64_t)h->nnz_per_row * h->n_dst;
  TmpBuf<int32_t> lohi;
  if ((rc = lohi.alloc(2, s))) return rc;   // (from the stream's block cache: a plain hipFree would synchronise the device)
  int32_t init[2] = {0x7fffffff, -1}, out[2];
  MPG_HIP(hipMemcpyAsync(lohi.p, init, sizeof(init), hipMemcpyHostToDevice, s));
  int gb = (int)((ni + 255) / 256);
  if (gb > 1024) gb = 1024;
  if (gb < 1) gb = 1;
  if (ni > 0) k_src_range<<<gb, 256, 0, s>>>(ni, ip, lohi.p);
  MPG_HIP(hipMemcpyAsync(out, lohi.p, sizeof(out), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  *first = out[1] >= 0 ? out[0] : 0;
  *end = out[1] >= 0 ? (int64_t)out[1] + 1 : 0;
  h->src_range_first = *first;
  h->src_range_end = *end;
  h->src_range_valid = true;
  return MPG_SUCCESS;
}

// keep_global: the shift of a mesh-wide source window (mpg_mesh_set_source_window): the handle stays what it was for
// mpg_handle_unique_sources / the Store cache; otherwise the handle becomes a re-indexed one (mpg_handle_rebase)
int mpg_k_rebase(mpg_handle_s *h, int64_t base, int64_t n_local, hipStream_t s, bool keep_global) {
  int rc;
  int32_t *ip = h->kind == MPG_KIND_CSR ? h->col.p : h->idx.p;
  int64_t ni = h->kind == MPG_KIND_CSR ? h->nnz : (int64_t)h->nnz_per_row * h->n_dst;
  TmpBuf<int32_t> bad;
  if ((rc = bad.alloc(1))) return rc;
  MPG_HIP(hipMemsetAsync(bad.p, 0, sizeof(int32_t), s));
  int gb = (int)((ni + 255) / 256);
  if (gb > 8192) gb = 8192;
  if (gb < 1) gb = 1;
  h->src_range_valid = false;   // the indices change
  k_rebase<<<gb, 256, 0, s>>>(ni, ip, (int32_t)base, (int32_t)n_local, bad.p);
  int32_t nb = 0;
  MPG_HIP(hipMemcpyAsync(&nb, bad.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  bad.free();
  h->n_src = n_local;
  if (!keep_global) h->localized = true;
  if (nb) {
    mpg_set_error("%d source indices fall outside [base, base+n_local)", nb);
    return MPG_ERR_INVALID_ARG;
  }
  return MPG_SUCCESS;
}

// mpg_init loads this translation unit's code object ahead of its first launch (mpg_api.hip: warm_modules)
const void *mpg_anchor_k_halo() { return (const void *)k_mark; }
