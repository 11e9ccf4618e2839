// Output epilogues: the arithmetic write_data.F90 does on rank 0 between ESMF_FieldGather and nf90_put_var
// (SURVEY s8(f) item 2), on the device-resident regridded fields, so only the final float32 arrays cross PCIe.
//
//   k_post_cast        dum3dt(:,:,:,1) = dum3d [- 300.0]   write_data.F90:1339-1347 (T), :1418 (PHB*9.81), and the
//                      float64 -> NF90_FLOAT conversion nf90_put_var applies to every field (:587-980 define
//                      all variables NF90_FLOAT)
//   k_post_layer_mean  Z_C(k) = 0.5*(PHB(k+1) + PHB(k))    write_data.F90:1406-1415
//   k_post_ptop        P_TOP = min(maxval(P_HYD), min over columns with P_HYD(top) >= 10 of 0.8*P_HYD(top))
//                                                           write_data.F90:1362-1371
// All are HBM streams (8 B read + 4 B written per value): 16-byte loads, one pass.
#include <string.h>

#include "geom.h"
#include "mpg_internal.h"

// dst_be: the float32 results are stored big-endian, as a NetCDF classic variable holds them (one v_perm_b32 per value)
template <bool VEC>
__global__ __launch_bounds__(256) void k_post_cast(const double *__restrict__ src, int64_t n, double scale, double offset,
                                                   float *__restrict__ dst, int dbe) {
  const Swz zd = make_swz(dbe);
  int64_t i = 2 * (blockIdx.x * (int64_t)blockDim.x + threadIdx.x);
  if (!VEC) {
    if (i < n) dst[i] = swz<true>((float)fma(src[i], scale, offset), zd);
    if (i + 1 < n) dst[i + 1] = swz<true>((float)fma(src[i + 1], scale, offset), zd);
  } else if (i + 1 < n) {
    double2 v = *reinterpret_cast<const double2 *>(src + i);
    float2 o = {swz<true>((float)fma(v.x, scale, offset), zd), swz<true>((float)fma(v.y, scale, offset), zd)};
    *reinterpret_cast<float2 *>(dst + i) = o;
  } else if (i < n) {
    dst[i] = swz<true>((float)fma(src[i], scale, offset), zd);
  }
}

// thread per target point, levels walked bottom-up with the previous level kept in a register: every source
// value is read once
__global__ __launch_bounds__(256) void k_post_layer_mean(const double *__restrict__ src, int nlevp1, int64_t P,
                                                         float *__restrict__ dst, int dbe) {
  const Swz zd = make_swz(dbe);
  int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p >= P) return;
  double prev = src[p];
  for (int k = 1; k < nlevp1; ++k) {
    double cur = src[(int64_t)k * P + p];
    dst[(int64_t)(k - 1) * P + p] = swz<true>((float)(0.5 * (cur + prev)), zd);
    prev = cur;
  }
}

// res[0] = max over the whole array, res[1] = min over the top level of 0.8*v where v >= 10 (else +inf).
// float64 atomics on ordered bit patterns are avoided: min/max are order independent, so a block reduction +
// one atomic per block on the integer image of the (non-negative or mixed-sign) doubles is exact.
__device__ __forceinline__ unsigned long long ord_key(double v) {
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
static double ord_val_host(unsigned long long k) {
  unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
  double v;
  memcpy(&v, &b, sizeof(v));
  return v;
}

__global__ __launch_bounds__(256) void k_post_ptop(const double *__restrict__ src, int nlev, int64_t P,
                                                   unsigned long long *__restrict__ keys) {
  __shared__ unsigned long long smax[256], smin[256];
  const int64_t n = (int64_t)nlev * P, top0 = (int64_t)(nlev - 1) * P;
  unsigned long long kmax = 0ull, kmin = ~0ull;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double v = src[i];
    unsigned long long k = ord_key(v);
    kmax = k > kmax ? k : kmax;
    if (i >= top0 && v >= 10.0) {
      unsigned long long k8 = ord_key(v * 0.8);
      kmin = k8 < kmin ? k8 : kmin;
    }
  }
  smax[threadIdx.x] = kmax;
  smin[threadIdx.x] = kmin;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      smax[threadIdx.x] = smax[threadIdx.x] > smax[threadIdx.x + st] ? smax[threadIdx.x] : smax[threadIdx.x + st];
      smin[threadIdx.x] = smin[threadIdx.x] < smin[threadIdx.x + st] ? smin[threadIdx.x] : smin[threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicMax(&keys[0], smax[0]);
    atomicMin(&keys[1], smin[0]);
  }
}

// in-place byte swap of n elements of 2, 4 or 8 bytes: NetCDF classic files are big-endian, so a variable can travel
// file -> device (and device -> file) as raw bytes and be turned around here at HBM speed instead of on a host core
template <typename T>
__global__ __launch_bounds__(256) void k_bswap(T *__restrict__ p, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    T v = p[i];
    if (sizeof(T) == 2) v = (T)(((v & 0xff) << 8) | ((v >> 8) & 0xff));
    else if (sizeof(T) == 4) v = (T)__builtin_bswap32((unsigned)v);
    else v = (T)__builtin_bswap64((unsigned long long)v);
    p[i] = v;
  }
}

int mpg_k_bswap(void *buf, int64_t n, int elem_size, hipStream_t s) {
  if (n == 0) return MPG_SUCCESS;
  unsigned nb = (unsigned)((n + 255) / 256);
  if (nb > 65536) nb = 65536;
  if (elem_size == 2) k_bswap<uint16_t><<<nb, 256, 0, s>>>((uint16_t *)buf, n);
  else if (elem_size == 4) k_bswap<uint32_t><<<nb, 256, 0, s>>>((uint32_t *)buf, n);
  else if (elem_size == 8) k_bswap<unsigned long long><<<nb, 256, 0, s>>>((unsigned long long *)buf, n);
  else {
    mpg_set_error("mpg_bswap_dev: element size must be 2, 4 or 8");
    return MPG_ERR_INVALID_ARG;
  }
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

int mpg_k_post_cast(const double *src, int64_t n, double scale, double offset, float *dst, int dst_be, hipStream_t s) {
  if (n == 0) return MPG_SUCCESS;
  int64_t nthr = (n + 1) / 2;
  bool vec = ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0;  // sub-arrays of a caller's tensor may be odd-aligned
  if (vec) k_post_cast<true><<<(unsigned)((nthr + 255) / 256), 256, 0, s>>>(src, n, scale, offset, dst, dst_be);
  else k_post_cast<false><<<(unsigned)((nthr + 255) / 256), 256, 0, s>>>(src, n, scale, offset, dst, dst_be);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

int mpg_k_post_layer_mean(const double *src, int nlevp1, int64_t P, float *dst, int dst_be, hipStream_t s) {
  if (P == 0 || nlevp1 < 2) return MPG_SUCCESS;
  k_post_layer_mean<<<(unsigned)((P + 255) / 256), 256, 0, s>>>(src, nlevp1, P, dst, dst_be);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

// the two reductions P_TOP is made of, for a host that holds only a block of the grid rows (one image per GPU): the
// blocks' maxima and candidate minima combine exactly (max of maxima, min of minima), the result does not depend on the split
int mpg_k_post_ptop_parts(const double *src, int nlev, int64_t P, double *vmax_host, double *candmin_host, int *has_cand_host, hipStream_t s) {
  TmpBuf<unsigned long long> keys;
  int rc;
  if ((rc = keys.alloc(2))) return rc;
  unsigned long long init[2] = {0ull, ~0ull}, out[2];
  MPG_HIP(hipMemcpyAsync(keys.p, init, sizeof(init), hipMemcpyHostToDevice, s));
  int64_t n = (int64_t)nlev * P;
  unsigned nb = (unsigned)((n + 255) / 256);
  if (nb > 4096) nb = 4096;
  k_post_ptop<<<nb, 256, 0, s>>>(src, nlev, P, keys.p);
  MPG_HIP(hipGetLastError());
  MPG_HIP(hipMemcpyAsync(out, keys.p, sizeof(out), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  *vmax_host = ord_val_host(out[0]);
  *has_cand_host = out[1] != ~0ull;
  *candmin_host = *has_cand_host ? ord_val_host(out[1]) : 0.0;
  return MPG_SUCCESS;
}

int mpg_k_post_ptop(const double *src, int nlev, int64_t P, double *ptop_host, hipStream_t s) {
  TmpBuf<unsigned long long> keys;
  int rc;
  if ((rc = keys.alloc(2))) return rc;
  unsigned long long init[2] = {0ull, ~0ull}, out[2];
  MPG_HIP(hipMemcpyAsync(keys.p, init, sizeof(init), hipMemcpyHostToDevice, s));
  int64_t n = (int64_t)nlev * P;
  unsigned nb = (unsigned)((n + 255) / 256);
  if (nb > 4096) nb = 4096;
  k_post_ptop<<<nb, 256, 0, s>>>(src, nlev, P, keys.p);
  MPG_HIP(hipGetLastError());
  MPG_HIP(hipMemcpyAsync(out, keys.p, sizeof(out), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  double vmax = ord_val_host(out[0]);
  double vmin = out[1] == ~0ull ? vmax : ord_val_host(out[1]);
  *ptop_host = vmin < vmax ? vmin : vmax;
  return MPG_SUCCESS;
}

// mpg_init loads this translation unit's code object ahead of its first launch (mpg_api.hip: warm_modules)
const void *mpg_anchor_k_post() { return (const void *)&k_post_cast<true>; }
