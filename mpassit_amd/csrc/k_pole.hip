// Pole caps of a periodic (monopole) source grid, Regrid side.
//
// ESMF_GridCreate1PeriDim(polekindflag=MONOPOLE) (model_grid.F90:685-694) closes each j end of the global
// lat-lon target grid with a pole node whose value is the mean of the neighbouring CENTER row.  The CENTER ->
// EDGE2 destaggering (interp.F90:316-327) therefore has destination points (the V rows at the poles) whose
// factor list holds the whole source row.  Those few points are not worth a CSR route for the whole handle:
// the handle stays 4-point (k_applyN<4>) and this kernel rewrites the cap points afterwards as
//   dst = sum_k w_k * src[idx_k]  (k_applyN's accumulation order)  +  w_pole * mean(row).
// One workgroup per (field, level): both row means are reduced in a fixed order (strided partial sums, then an
// LDS tree), so the result does not depend on scheduling.
#include "geom.h"
#include "mpg_internal.h"

template <typename TS, typename TD, bool SWZ>
__global__ __launch_bounds__(256) void k_pole_fix(const int32_t *__restrict__ idx, const double *__restrict__ w,
                                                  const int32_t *__restrict__ pole_dst, const int32_t *__restrict__ pole_src0,
                                                  const double *__restrict__ pole_w, int n_pole, int row_len,
                                                  const TS *__restrict__ src, TD *__restrict__ dst, int64_t P, int64_t nsrc, int nlev,
                                                  int lev_fast, double scale, double offset, int sbe, int dbe, FieldTab tab) {
  __shared__ double red[2][256];
  const Swz zs = make_swz(sbe), zd = make_swz(dbe);
  const int k = blockIdx.x % nlev, fld = blockIdx.x / nlev;
  const TS *sf = mpg_field_src(tab, src, fld, (int64_t)nlev * nsrc);
  TD *df = mpg_field_dst(tab, dst, fld, (int64_t)nlev * P);
  offset = mpg_field_off(tab, fld, offset);
  const int64_t row1 = nsrc - row_len;  // first source of the last CENTER row
  double s0 = 0.0, s1 = 0.0;
  for (int i = threadIdx.x; i < row_len; i += 256) {
    int64_t c0 = i, c1 = row1 + i;
    s0 += (double)swz<SWZ>(lev_fast ? sf[c0 * nlev + k] : sf[(int64_t)k * nsrc + c0], zs);
    s1 += (double)swz<SWZ>(lev_fast ? sf[c1 * nlev + k] : sf[(int64_t)k * nsrc + c1], zs);
  }
  red[0][threadIdx.x] = s0;
  red[1][threadIdx.x] = s1;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      red[0][threadIdx.x] += red[0][threadIdx.x + st];
      red[1][threadIdx.x] += red[1][threadIdx.x + st];
    }
    __syncthreads();
  }
  const double mean0 = red[0][0] / (double)row_len, mean1 = red[1][0] / (double)row_len;
  for (int q = threadIdx.x; q < n_pole; q += 256) {
    double wp = pole_w[q];
    if (wp == 0.0) continue;
    int64_t p = pole_dst[q];
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int64_t c = idx[j * P + p];
      acc = fma(w[j * P + p], (double)swz<SWZ>(lev_fast ? sf[c * nlev + k] : sf[(int64_t)k * nsrc + c], zs), acc);
    }
    acc = fma(wp, pole_src0[q] == 0 ? mean0 : mean1, acc);
    df[(int64_t)k * P + p] = swz<SWZ>((TD)fma(acc, scale, offset), zd);
  }
}

template <typename TS, typename TD>
static int launch_pole(mpg_handle_s *h, const void *src, int layout, int nlev, int nfields, void *dst, double scale, double offset, int sbe, int dbe,
                       hipStream_t s, const FieldTab &tab) {
  auto fn = (sbe || dbe) ? k_pole_fix<TS, TD, true> : k_pole_fix<TS, TD, false>;
  fn<<<(unsigned)(nlev * nfields), 256, 0, s>>>(h->idx.p, h->w.p, h->pole_dst.p, h->pole_src0.p, h->pole_w.p, (int)h->n_pole, h->pole_len,
                                               (const TS *)src, (TD *)dst, h->n_dst, h->n_src, nlev, layout == MPG_LAYOUT_LEV_FAST, scale, offset, sbe,
                                               dbe, tab);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

int mpg_k_pole_fix(mpg_handle_s *h, const void *src, int src_type, int layout, int nlev, int nfields, void *dst, int dst_type,
                   double scale, double offset, hipStream_t s, const FieldTab &tab) {
  if (h->n_pole == 0 || nlev == 0 || nfields == 0) return MPG_SUCCESS;
  if (h->kind != MPG_KIND_FIXED || h->nnz_per_row != 4 || h->pole_len <= 0 || h->pole_len > h->n_src) {
    mpg_set_error("pole terms on a handle that is not a Grid -> Grid bilinear one");
    return MPG_ERR_INVALID_ARG;
  }
  const int sf32 = src_type & MPG_TYPE_F32, df32 = dst_type & MPG_TYPE_F32, sbe = (src_type & MPG_TYPE_BE) != 0, dbe = (dst_type & MPG_TYPE_BE) != 0;
  if (sf32 && df32) return launch_pole<float, float>(h, src, layout, nlev, nfields, dst, scale, offset, sbe, dbe, s, tab);
  if (sf32) return launch_pole<float, double>(h, src, layout, nlev, nfields, dst, scale, offset, sbe, dbe, s, tab);
  if (df32) return launch_pole<double, float>(h, src, layout, nlev, nfields, dst, scale, offset, sbe, dbe, s, tab);
  return launch_pole<double, double>(h, src, layout, nlev, nfields, dst, scale, offset, sbe, dbe, s, tab);
}

// mpg_init loads this translation unit's code object ahead of its first launch (mpg_api.hip: warm_modules)
const void *mpg_anchor_k_pole() { return (const void *)&k_pole_fix<double, double, false>; }
