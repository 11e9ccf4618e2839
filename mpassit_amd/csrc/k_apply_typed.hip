// Fused ingest / egress variants of the Regrid kernels (the callers either side of the
// hot path).  The reference widens the single-precision MPAS history variables to float64 when it reads them
// (nf90_get_var into real(8) buffers, input_data.F90:630-655), regrids in float64 and narrows every output
// variable back to float32 when it writes (all NF90_FLOAT, write_data.F90:779), after two scalar post-ops:
// T - 300 (write_data.F90:1343) and PHB * 9.81 (:1418).  These kernels read float32 or float64 sources, do the
// identical float64 arithmetic and store float32 or float64 with an affine epilogue:
//     dst = (TD)( regrid(src) * scale + offset )
// so the float32 results are bit-identical to what the reference's writer would put in the file, while the
// HBM traffic per 3-D field drops from 8+8 to 4+4 bytes per source/destination element.
// k_apply3_cf_t / k_apply3_lf_t / k_apply3_lf_f32x2 have the structure of k_apply3_cf<2,4,true> / k_apply3_lf<64> in k_apply.hip;
// k_apply3_lf_f32m (further down) is the level-fast row gather both entry points use by default, for float32 and float64 rows.
#include <algorithm>

#include "geom.h"
#include "mpg_internal.h"

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void k_apply3_cf_t(const int32_t *__restrict__ idx, const double *__restrict__ w,
                                                     const TS *__restrict__ src, TD *__restrict__ dst, int nx, int ny, int64_t nsrc,
                                                     int nlev, int ntx, int nty, double scale, double offset) {
  constexpr int RPT = 2, TY = 4 * RPT;
  int64_t P = (int64_t)nx * ny;
  unsigned ntile = (unsigned)ntx * nty;
  unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  unsigned tile = lin % ntile;
  int fld = lin / ntile;
  int tx = tile % ntx, ty = tile / ntx;
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int j0 = ty * TY + wave * RPT;
  int32_t c[RPT][3];
  double ww[RPT][3];
  bool act[RPT], mapped[RPT];
  int64_t po[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    int j = j0 + r;
    int i = tx * 64 + lane - mpg_tile_shift(j, nx);   // row-shifted tile: aligned store segments (mpg_internal.h)
    act[r] = (i >= 0) && (i < nx) && (j < ny);
    int64_t p = act[r] ? (int64_t)j * nx + i : 0;
    po[r] = p;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      c[r][q] = idx[q * P + p];
      ww[r][q] = w[q * P + p];
    }
    mapped[r] = c[r][0] >= 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) c[r][q] = max(c[r][q], 0);
  }
  const TS *s = src + (int64_t)fld * nlev * nsrc;
  TD *d = dst + (int64_t)fld * nlev * P;
  for (int k = 0; k < nlev; ++k) {
    __syncthreads();
    double v[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      double a = (double)s[c[r][0]], b = (double)s[c[r][1]], e = (double)s[c[r][2]];
      v[r] = wsum3(ww[r][0], a, ww[r][1], b, ww[r][2], e);
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r)
      if (act[r]) __builtin_nontemporal_store((TD)fma(mapped[r] ? v[r] : 0.0, scale, offset), d + po[r]);
    s += nsrc;
    d += P;
  }
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void k_apply3_lf_t(const int32_t *__restrict__ idx, const double *__restrict__ w,
                                                     const TS *__restrict__ src, TD *__restrict__ dst, int nx, int ny, int64_t nsrc,
                                                     int nlev, int ntx, int nty, double scale, double offset) {
  extern __shared__ double sw[];    // sw[3][64] | sidx[3][64] | tile[nlev][65] in the DESTINATION type (narrowing at the tile
  int32_t *sidx = (int32_t *)(sw + 192);            // write or at the store gives the same bits; float32 halves the LDS -> 8 WGs / CU)
  TD *tile = (TD *)(sidx + 192);
  int64_t P = (int64_t)nx * ny;
  unsigned ntile = (unsigned)ntx * nty;
  unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  unsigned tl = lin % ntile;
  int fld = lin / ntile;
  int tx = tl % ntx, ty = tl / ntx;
  int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < 192) {
    int pt = t & 63, q = t >> 6;
    int j = ty, i = tx * 64 + pt - mpg_tile_shift(j, nx);
    bool in = i >= 0 && i < nx && j < ny;
    int64_t p = in ? (int64_t)j * nx + i : 0;
    int32_t c = idx[q * P + p];
    sidx[q * 64 + pt] = in ? c : -1;
    sw[q * 64 + pt] = w[q * P + p];
  }
  __syncthreads();
  const TS *sf = src + (int64_t)fld * nlev * nsrc;
  for (int kb = 0; kb < nlev; kb += 64) {
    int k = kb + lane;
    bool kact = k < nlev;
    int kk = kact ? k : 0;
#pragma unroll
    for (int q0 = 0; q0 < 16; q0 += 4) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int pt = wave * 16 + q0 + u;
        int32_t c0 = __builtin_amdgcn_readfirstlane(sidx[pt]);
        int32_t c1 = __builtin_amdgcn_readfirstlane(sidx[64 + pt]);
        int32_t c2 = __builtin_amdgcn_readfirstlane(sidx[128 + pt]);
        double w0 = sw[pt], w1 = sw[64 + pt], w2 = sw[128 + pt];
        bool m = c0 >= 0;
        c0 = max(c0, 0); c1 = max(c1, 0); c2 = max(c2, 0);
        const TS *r0 = sf + (int64_t)c0 * nlev, *r1 = sf + (int64_t)c1 * nlev, *r2 = sf + (int64_t)c2 * nlev;
        double a = (double)r0[kk], b = (double)r1[kk], e = (double)r2[kk];
        v[u] = fma(m ? wsum3(w0, a, w1, b, w2, e) : 0.0, scale, offset);
      }
      if (kact) {
#pragma unroll
        for (int u = 0; u < 4; ++u) tile[k * 65 + wave * 16 + q0 + u] = (TD)v[u];
      }
    }
  }
  __syncthreads();
  TD *df = dst + (int64_t)fld * nlev * P;
  int j = ty, i = tx * 64 + lane - mpg_tile_shift(j, nx);
  if (i >= 0 && i < nx && j < ny) {
    int64_t p = (int64_t)j * nx + i;
    for (int k = wave; k < nlev; k += 4) __builtin_nontemporal_store(tile[k * 65 + lane], df + (int64_t)k * P + p);
  }
}

// float32 rows in file order, two levels per lane: a half-wave covers 64 levels of one point with ONE 8-byte load per
// row (rows are only 4-byte aligned -- 55 levels = 220 bytes -- which the hardware's unaligned access mode takes), so a
// wavefront works on two points at a time and issues half the load instructions of k_apply3_lf_t, which is bound by
// its instruction / latency budget, not by bytes, once the elements are 4 bytes (C4: 2.9 TB/s -> 3.8).  The lane that
// would read past the end of a row reads the row's last two levels instead and shifts.  Same tile, same transposing
// store, same wsum3 arithmetic -> bit-identical results.  Needs nlev >= 2.  (Four levels per lane with 16-byte loads
// was measured too: slower, 14 of 16 lanes busy at 55 levels and more select / LDS work per load.)
typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));

template <typename TD, int WPE, int UNR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void k_apply3_lf_f32x2(const int32_t *__restrict__ idx, const double *__restrict__ w,
                                                         const float *__restrict__ src, TD *__restrict__ dst, int nx, int ny, int64_t nsrc,
                                                         int nlev, int ntx, int nty, double scale, double offset, unsigned band) {
  extern __shared__ double sw[];    // sw[3][64] | sidx[3][64] | tile[nlev][65] in the DESTINATION type (narrowing at the tile
  int32_t *sidx = (int32_t *)(sw + 192);            // write or at the store gives the same bits; float32 halves the LDS -> 8 WGs / CU)
  TD *tile = (TD *)(sidx + 192);
  int64_t P = (int64_t)nx * ny;
  unsigned ntile = (unsigned)ntx * nty;
  unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  unsigned tl = band_order(lin % ntile, ntx, nty, band);
  int fld = lin / ntile;
  int tx = tl % ntx, ty = tl / ntx;
  int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < 192) {
    int pt = t & 63, q = t >> 6;
    int i = tx * 64 + pt, j = ty;
    bool in = i < nx && j < ny;
    int64_t p = in ? (int64_t)j * nx + i : 0;
    int32_t c = idx[q * P + p];
    sidx[q * 64 + pt] = in ? c : -1;
    sw[q * 64 + pt] = w[q * P + p];
  }
  __syncthreads();
  const float *sf = src + (int64_t)fld * nlev * nsrc;
  const int half = lane >> 5, sl = lane & 31;
  for (int kb = 0; kb < nlev; kb += 64) {
    const int k0 = kb + 2 * sl;
    const int base = min(k0, nlev - 2);            // base < k0 only on the lane that holds the end of the row
    const bool shifted = base != k0, a0 = k0 < nlev, a1 = k0 + 1 < nlev;
#pragma unroll UNR
    for (int q0 = 0; q0 < 16; q0 += 4) {           // two pairs of points per step: 6 row loads in flight per lane
      double v[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int pt = wave * 16 + q0 + 2 * u + half;
        int32_t c0 = sidx[pt], c1 = sidx[64 + pt], c2 = sidx[128 + pt];
        const double w0 = sw[pt], w1 = sw[64 + pt], w2 = sw[128 + pt];
        const bool m = c0 >= 0;
        c0 = max(c0, 0); c1 = max(c1, 0); c2 = max(c2, 0);
        const f32x2_u x0 = *(const f32x2_u *)(sf + (int64_t)c0 * nlev + base);
        const f32x2_u x1 = *(const f32x2_u *)(sf + (int64_t)c1 * nlev + base);
        const f32x2_u x2 = *(const f32x2_u *)(sf + (int64_t)c2 * nlev + base);
        const double a = shifted ? x0.y : x0.x, b = shifted ? x1.y : x1.x, e = shifted ? x2.y : x2.x;
        v[u][0] = fma(m ? wsum3(w0, a, w1, b, w2, e) : 0.0, scale, offset);
        v[u][1] = fma(m ? wsum3(w0, (double)x0.y, w1, (double)x1.y, w2, (double)x2.y) : 0.0, scale, offset);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int pt = wave * 16 + q0 + 2 * u + half;
        if (a0) tile[k0 * 65 + pt] = (TD)v[u][0];
        if (a1) tile[(k0 + 1) * 65 + pt] = (TD)v[u][1];
      }
    }
  }
  __syncthreads();
  TD *df = dst + (int64_t)fld * nlev * P;
  int i = tx * 64 + lane, j = ty;
  if (i < nx && j < ny) {
    int64_t p = (int64_t)j * nx + i;
    for (int k = wave; k < nlev; k += 4) __builtin_nontemporal_store(tile[k * 65 + lane], df + (int64_t)k * P + p);
  }
}

// The same row gather with LINEAR tiles and several fields per workgroup.  A tile is 64 consecutive target points of the
// flattened [ny][nx] plane starting at a multiple of 64, so every store of a level is one naturally aligned 256-byte
// (float32) / 512-byte (float64) segment whatever nx is (64 x 1 tiles of a 1800-wide grid start 32 bytes off a line in three
// rows of four: 10 % more bytes written, PMC, and two partial lines per store); the three cell offsets (premultiplied by
// nlev, 32 bit, added to a scalar field base: no 64-bit address arithmetic per load) and weights of the tile are fetched
// ONCE and serve `fpw` fields of the bundle (per field they are 36 of the 542 bytes a point moves at 55 float32 levels).
// TS = float64 rows work the same way with one 16-byte load per lane (rows are 8-byte aligned); EPI = false leaves the
// affine epilogue out (mpg_regrid_dev: the float64 result as it stands, sign of zero included).
typedef double f64x2_u __attribute__((ext_vector_type(2), aligned(8)));
template <typename TS> struct Row2;
template <> struct Row2<float> { typedef f32x2_u type; };
template <> struct Row2<double> { typedef f64x2_u type; };

template <typename TS, typename TD, int UNR, int WPE, bool EPI = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void k_apply3_lf_f32m(
    const int32_t *__restrict__ idx, const double *__restrict__ w, const TS *__restrict__ src, TD *__restrict__ dst, int64_t P, int64_t nsrc,
    int nlev, unsigned ntile, int nfields, int fpw, double scale, double offset) {
  typedef typename Row2<TS>::type row2;
  extern __shared__ double sw[];                    // sw[3][64] | soff[3][64] | tile[nlev][65] in the destination type
  uint32_t *soff = (uint32_t *)(sw + 192);
  TD *tile = (TD *)(soff + 192);
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned tl = lin % ntile;
  const int f0 = (int)(lin / ntile) * fpw, f1 = min(nfields, f0 + fpw);
  const int64_t p0 = (int64_t)tl * 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < 192) {
    const int pt = t & 63, q = t >> 6;
    const bool in = p0 + pt < P;
    const int64_t p = in ? p0 + pt : 0;
    const int32_t c = idx[q * P + p];
    soff[q * 64 + pt] = (in && c >= 0) ? (uint32_t)c * (uint32_t)nlev : 0xFFFFFFFFu;
    sw[q * 64 + pt] = w[q * P + p];
  }
  __syncthreads();
  const int half = lane >> 5, sl = lane & 31;
  const bool store_lane = p0 + lane < P;
  for (int f = f0; f < f1; ++f) {
    const TS *sf = src + (int64_t)f * nlev * nsrc;
    for (int kb = 0; kb < nlev; kb += 64) {
      // Branch-free body (a divergent `mapped ?` / `level < nlev ?` made the compiler wait for each point's three loads before
      // it issued the next point's): lanes past the end of a row are clamped onto its last two levels and re-write the
      // values of level nlev-1 where the lane that owns that level writes the same bits; unmapped points read row 0 and
      // their result is masked to +0.0 before the epilogue.
      const int k0 = kb + 2 * sl;
      const int base = min(k0, nlev - 2);
      const bool shifted = base != k0;
      const int kw0 = min(k0, nlev - 1) * 65, kw1 = min(k0 + 1, nlev - 1) * 65;
#pragma unroll UNR
      for (int q0 = 0; q0 < 16; q0 += 4) {
        row2 x[2][3];
        double ww[2][3];
        unsigned long long keep[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int pt = wave * 16 + q0 + 2 * u + half;
          uint32_t o0 = soff[pt], o1 = soff[64 + pt], o2 = soff[128 + pt];
          ww[u][0] = sw[pt]; ww[u][1] = sw[64 + pt]; ww[u][2] = sw[128 + pt];
          const bool m = o0 != 0xFFFFFFFFu;
          keep[u] = m ? ~0ull : 0ull;
          o0 = m ? o0 : 0u; o1 = m ? o1 : 0u; o2 = m ? o2 : 0u;
          x[u][0] = *(const row2 *)(sf + (o0 + (uint32_t)base));
          x[u][1] = *(const row2 *)(sf + (o1 + (uint32_t)base));
          x[u][2] = *(const row2 *)(sf + (o2 + (uint32_t)base));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int pt = wave * 16 + q0 + 2 * u + half;
          const double a = shifted ? x[u][0].y : x[u][0].x, b = shifted ? x[u][1].y : x[u][1].x, e = shifted ? x[u][2].y : x[u][2].x;
          const double r0 = wsum3(ww[u][0], a, ww[u][1], b, ww[u][2], e);
          const double r1 = wsum3(ww[u][0], (double)x[u][0].y, ww[u][1], (double)x[u][1].y, ww[u][2], (double)x[u][2].y);
          double v0 = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(r0) & keep[u]));
          double v1 = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(r1) & keep[u]));
          if constexpr (EPI) {
            v0 = fma(v0, scale, offset);
            v1 = fma(v1, scale, offset);
          }
          tile[kw0 + pt] = (TD)v0;
          tile[kw1 + pt] = (TD)v1;
        }
      }
    }
    __syncthreads();
    TD *df = dst + (int64_t)f * nlev * P + p0;
    if (store_lane)
      for (int k = wave; k < nlev; k += 4) __builtin_nontemporal_store(tile[k * 65 + lane], df + (int64_t)k * P + lane);
    __syncthreads();                                 // the tile is rewritten by the next field
  }
}

// nearest (NNZ = 1, weights implicit), 4-point destagger (NNZ = 4) and CSR (NNZ = 0): one thread per target point
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void k_apply_generic_t(int nnz_per_row, const int32_t *__restrict__ idx, const double *__restrict__ w,
                                                         const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                         const double *__restrict__ val, const TS *__restrict__ src,
                                                         TD *__restrict__ dst, int64_t P, int64_t nsrc, int nlev, int lev_fast, int nblk,
                                                         double scale, double offset) {
  unsigned blk = blockIdx.x % nblk;
  int fld = blockIdx.x / nblk;
  int64_t p = (int64_t)blk * 256 + threadIdx.x;
  if (p >= P) return;
  const TS *sf = src + (int64_t)fld * nlev * nsrc;
  TD *df = dst + (int64_t)fld * nlev * P;
  for (int k = 0; k < nlev; ++k) {
    double acc = 0.0;
    if (nnz_per_row == 0) {
      for (int q = rowptr[p]; q < rowptr[p + 1]; ++q) {
        int32_t c = col[q];
        acc = fma(val[q], (double)(lev_fast ? sf[(int64_t)c * nlev + k] : sf[(int64_t)k * nsrc + c]), acc);
      }
    } else if (nnz_per_row == 1) {
      int32_t c = idx[p];
      if (c >= 0) acc = (double)(lev_fast ? sf[(int64_t)c * nlev + k] : sf[(int64_t)k * nsrc + c]);
    } else if (idx[p] >= 0) {
      for (int q = 0; q < nnz_per_row; ++q) {
        int32_t c = idx[q * P + p];
        acc = fma(w[q * P + p], (double)(lev_fast ? sf[(int64_t)c * nlev + k] : sf[(int64_t)k * nsrc + c]), acc);
      }
    }
    df[(int64_t)k * P + p] = (TD)fma(acc, scale, offset);
  }
}

template <typename TS, typename TD>
static int launch_typed(mpg_handle_s *h, const void *src, int layout, int nlev, int nfields, void *dst, double scale, double offset,
                        hipStream_t s) {
  int64_t P = h->n_dst;
  int lev_fast = layout == MPG_LAYOUT_LEV_FAST;
  if (h->kind == MPG_KIND_FIXED && h->nnz_per_row == 3) {
    if (lev_fast) {
      int ntx = (h->nx_dst + 63) / 64, nty = h->ny_dst;
      size_t lds = sizeof(TD) * 65 * (size_t)nlev + sizeof(double) * 192 + sizeof(int32_t) * 192;
      if (lds > 160 * 1024) {
        mpg_set_error("Regrid(LEV_FAST): %d levels exceed the LDS tile", nlev);
        return MPG_ERR_UNSUPPORTED;
      }
      // default: linear 64-point tiles, 32-bit row offsets, two levels per lane (k_apply3_lf_f32m, float32 and float64 rows);
      // lf_variant 4 / 401-403 keep the 64 x 1 tiles of a grid row for comparison, 410-418 are the measured alternatives
      // (fields per workgroup, unroll, occupancy)
      const int lfv = mpg_lf_variant();
      if (nlev >= 2 && !(lfv == 4 || (lfv >= 401 && lfv <= 403)) && (uint64_t)h->n_src * (uint64_t)nlev < 0xFFFFFFFFull) {
        static const int fpws[] = {1, 2, 4, 1 << 20, 13, 7, 1, 1, 1, 1};
        const int v = (lfv >= 410 && lfv < 420) ? lfv - 410 : (sizeof(TS) == 4 ? 6 : 0);   // measured: unroll 2 for float32 rows, 1 for float64
        const int fpw = std::min(nfields, fpws[v]);
        const unsigned ntile = (unsigned)((P + 63) / 64), ngroups = (unsigned)((nfields + fpw - 1) / fpw);
        auto fn = k_apply3_lf_f32m<TS, TD, 1, 8>;
        if (v == 6) fn = k_apply3_lf_f32m<TS, TD, 2, 8>;
        if (v == 7) fn = k_apply3_lf_f32m<TS, TD, 4, 4>;
        if (v == 8) fn = k_apply3_lf_f32m<TS, TD, 2, 6>;
        if (lds > 48 * 1024) MPG_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        fn<<<ntile * ngroups, 256, lds, s>>>(h->idx.p, h->w.p, (const TS *)src, (TD *)dst, P, h->n_src, nlev, ntile, nfields, fpw, scale, offset);
        MPG_HIP(hipGetLastError());
        return MPG_SUCCESS;
      }
      if (sizeof(TS) == 4 && nlev >= 2) {   // float32 rows: two levels per lane, two points per wavefront pass
        auto go = [&](auto fn) -> int {
          if (lds > 48 * 1024) MPG_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          fn<<<(unsigned)ntx * nty * nfields, 256, lds, s>>>(h->idx.p, h->w.p, (const float *)src, (TD *)dst, h->nx_dst, h->ny_dst, h->n_src, nlev,
                                                            ntx, nty, scale, offset, (unsigned)mpg_tile_band());
          return MPG_SUCCESS;
        };
        int rc0;
        switch (mpg_lf_variant()) {   // 401-403: occupancy experiments (registers capped for 5 / 6 / 8 waves per SIMD)
          case 401: rc0 = go(k_apply3_lf_f32x2<TD, 8, 1>); break;
          case 402: rc0 = go(k_apply3_lf_f32x2<TD, 5, 2>); break;
          case 403: rc0 = go(k_apply3_lf_f32x2<TD, 8, 2>); break;
          default: rc0 = go(k_apply3_lf_f32x2<TD, 4, 4>); break;
        }
        if (rc0) return rc0;
      } else {
        if (lds > 48 * 1024)
          MPG_HIP(hipFuncSetAttribute((const void *)k_apply3_lf_t<TS, TD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int ntxs = mpg_tile_ntx(h->nx_dst, 64);
        k_apply3_lf_t<TS, TD><<<(unsigned)ntxs * nty * nfields, 256, lds, s>>>(h->idx.p, h->w.p, (const TS *)src, (TD *)dst, h->nx_dst,
                                                                              h->ny_dst, h->n_src, nlev, ntxs, nty, scale, offset);
      }
    } else {
      int ntx = mpg_tile_ntx(h->nx_dst, 64), nty = (h->ny_dst + 7) / 8;
      k_apply3_cf_t<TS, TD><<<(unsigned)ntx * nty * nfields, 256, 0, s>>>(h->idx.p, h->w.p, (const TS *)src, (TD *)dst, h->nx_dst,
                                                                         h->ny_dst, h->n_src, nlev, ntx, nty, scale, offset);
    }
  } else {
    int nblk = (int)((P + 255) / 256);
    k_apply_generic_t<TS, TD><<<(unsigned)nblk * nfields, 256, 0, s>>>(h->kind == MPG_KIND_CSR ? 0 : h->nnz_per_row, h->idx.p, h->w.p,
                                                                      h->rowptr.p, h->col.p, h->val.p, (const TS *)src, (TD *)dst, P,
                                                                      h->n_src, nlev, lev_fast, nblk, scale, offset);
  }
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

// float64 rows in, float64 out, no epilogue: the level-fast row gather of mpg_regrid_dev (k_apply.hip).
// -> MPG_ERR_UNSUPPORTED when the 32-bit row offsets or the LDS tile do not fit (the caller keeps its older kernel).
int mpg_k_apply3_lf_rows(mpg_handle_s *h, const double *src, int nlev, int nfields, double *dst, hipStream_t s) {
  const int64_t P = h->n_dst;
  const size_t lds = sizeof(double) * 65 * (size_t)nlev + sizeof(double) * 192 + sizeof(int32_t) * 192;
  if (nlev < 2 || (uint64_t)h->n_src * (uint64_t)nlev >= 0xFFFFFFFFull || lds > 160 * 1024) return MPG_ERR_UNSUPPORTED;
  const unsigned ntile = (unsigned)((P + 63) / 64);
  auto fn = k_apply3_lf_f32m<double, double, 1, 8, false>;
  if (lds > 48 * 1024) MPG_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  fn<<<ntile * (unsigned)nfields, 256, lds, s>>>(h->idx.p, h->w.p, src, dst, P, h->n_src, nlev, ntile, nfields, 1, 1.0, 0.0);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

int mpg_k_apply_typed(mpg_handle_s *h, const void *src, int src_f32, int layout, int nlev, int nfields, void *dst, int dst_f32,
                      double scale, double offset, hipStream_t s) {
  if (h->n_dst == 0 || nlev == 0 || nfields == 0) return MPG_SUCCESS;
  if (h->n_src == 0) {  // nothing mapped: the destination is the epilogue of 0.0
    if (offset != 0.0) {
      mpg_set_error("mpg_regrid_typed: handle without sources and a non-zero offset is not supported");
      return MPG_ERR_UNSUPPORTED;
    }
    MPG_HIP(hipMemsetAsync(dst, 0, (dst_f32 ? 4 : 8) * (size_t)h->n_dst * nlev * nfields, s));
    return MPG_SUCCESS;
  }
  if (h->kind == MPG_KIND_FIXED && h->nnz_per_row == 3 && layout == MPG_LAYOUT_CELL_FAST && mpg_a3_staged() != -2 && !h->cft_unfit) {
    int rc = mpg_k_apply3_cfu_typed(h, src, src_f32, nlev, nfields, dst, dst_f32, scale, offset, s);  // LDS-staged (k_apply_lfu.hip)
    if (rc != MPG_ERR_UNSUPPORTED) return rc;
    h->cft_unfit = true;  // tile lists too long for the staged typed kernel: lane-gather from now on (decided once per handle)
  }
  if (h->kind == MPG_KIND_FIXED && h->nnz_per_row == 3 && layout == MPG_LAYOUT_LEV_FAST && mpg_lf_variant() == 200) {
    int rc = mpg_k_apply3_lfr(h, src, src_f32, nlev, nfields, dst, dst_f32, scale, offset, s);
    if (rc != MPG_ERR_UNSUPPORTED) return rc;
  }
  if (h->kind == MPG_KIND_FIXED && h->nnz_per_row == 3 && layout == MPG_LAYOUT_LEV_FAST && mpg_lf_variant() >= 300 && mpg_lf_variant() < 400) {
    int rc = mpg_k_apply3_lfs(h, mpg_lf_variant() - 300, src, src_f32, nlev, nfields, dst, dst_f32, scale, offset, (size_t)160 * 1024, s);
    if (rc == MPG_SUCCESS && h->n_pole) rc = mpg_k_pole_fix(h, src, src_f32, layout, nlev, nfields, dst, dst_f32, scale, offset, s);
    if (rc != MPG_ERR_UNSUPPORTED) return rc;
  }
  if (h->kind == MPG_KIND_FIXED && h->nnz_per_row == 3 && layout == MPG_LAYOUT_LEV_FAST && mpg_lf_variant() == -1) {
    int rc = mpg_k_apply3_lfu_typed(h, src, src_f32, nlev, nfields, dst, dst_f32, scale, offset, s);  // staged, when it pays
    if (rc != MPG_ERR_UNSUPPORTED) return rc;
  }
  int rc;
  if (src_f32 && dst_f32) rc = launch_typed<float, float>(h, src, layout, nlev, nfields, dst, scale, offset, s);
  else if (src_f32) rc = launch_typed<float, double>(h, src, layout, nlev, nfields, dst, scale, offset, s);
  else if (dst_f32) rc = launch_typed<double, float>(h, src, layout, nlev, nfields, dst, scale, offset, s);
  else rc = launch_typed<double, double>(h, src, layout, nlev, nfields, dst, scale, offset, s);
  if (rc == MPG_SUCCESS && h->n_pole) rc = mpg_k_pole_fix(h, src, src_f32, layout, nlev, nfields, dst, dst_f32, scale, offset, s);
  return rc;
}
