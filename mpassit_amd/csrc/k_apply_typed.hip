// Fused ingest / egress variants of the Regrid kernels (the callers either side of the
// hot path).  The reference widens the single-precision MPAS history variables to float64 when it reads them
// (nf90_get_var into real(8) buffers, input_data.F90:630-655), regrids in float64 and narrows every output
// variable back to float32 when it writes (all NF90_FLOAT, write_data.F90:779), after two scalar post-ops:
// T - 300 (write_data.F90:1343) and PHB * 9.81 (:1418).  These kernels read float32 or float64 sources, do the
// identical float64 arithmetic and store float32 or float64 with an affine epilogue:
//     dst = (TD)( regrid(src) * scale + offset )
// so the float32 results are bit-identical to what the reference's writer would put in the file, while the
// HBM traffic per 3-D field drops from 8+8 to 4+4 bytes per source/destination element.  Either side may be
// big-endian (MPG_TYPE_BE): a NetCDF classic variable is consumed and produced as the file stores it, the byte
// reversal is a v_perm_b32 in the load / store path (geom.h swz), no swap pass exists.
//   k_apply3_cf_t     lane gather, cell-fast source (structure of k_apply3_cf, k_apply.hip)
//   k_apply3_lf_rows  the level-fast row gather on LINEAR tiles -- the default of both entry points for file-order
//                     sources whose target points share few cells (configuration 4), float32 and float64 rows
//   k_apply3_lf_t     row gather on grid-row tiles: the capacity fallback (structure of k_apply3_lf)
//   k_apply_generic_t nearest / 4-point destagger / CSR, one thread per target point
#include <algorithm>

#include "geom.h"
#include "mpg_internal.h"

template <typename TS, typename TD, bool SWZ>
__global__ __launch_bounds__(256) void k_apply3_cf_t(const int32_t *__restrict__ idx, const double *__restrict__ w,
                                                     const TS *__restrict__ src, TD *__restrict__ dst, int nx, int ny, int64_t nsrc,
                                                     int nlev, int ntx, int nty, double scale, double offset, int sbe, int dbe, FieldTab tab) {
  constexpr int RPT = 2, TY = 4 * RPT;
  const Swz zs = make_swz(sbe), zd = make_swz(dbe);
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
  const TS *s = mpg_field_src(tab, src, fld, (int64_t)nlev * nsrc);
  TD *d = mpg_field_dst(tab, dst, fld, (int64_t)nlev * P);
  offset = mpg_field_off(tab, fld, offset);
  for (int k = 0; k < nlev; ++k) {
    __syncthreads();
    double v[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      double a = (double)swz<SWZ>(s[c[r][0]], zs), b = (double)swz<SWZ>(s[c[r][1]], zs), e = (double)swz<SWZ>(s[c[r][2]], zs);
      v[r] = wsum3(ww[r][0], a, ww[r][1], b, ww[r][2], e);
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r)
      if (act[r]) stream_store_lane(swz<SWZ>((TD)fma(mapped[r] ? v[r] : 0.0, scale, offset), zd), d + po[r], (unsigned)lane * (unsigned)sizeof(TD));   // geom.h: per lane
    s += nsrc;
    d += P;
  }
}

// A bundle of 2-D fields held in SEPARATE arrays (interp.F90:123-134 diag bundle, :207-219 hist 2-D bundle; mpg_regrid_bundle_typed_dev
// with nlev = 1): the fields take the place of the levels -- a workgroup keeps its 64 x 8 points' indices and weights in registers and
// walks the bundle's fields, where one launch item per (field, tile) would load those 36 bytes per point once per FIELD to produce 8
// (the staged kernel did, after building tile lists no 3-D Regrid of a file-order job ever uses: 0.47 + 0.95 ms of a cold
// configuration-4 job for its 19 diag fields, round 6 timeline).  Arithmetic and epilogue of k_apply3_cf_t: the same bits.
template <typename TS, typename TD, bool SWZ>
__global__ __launch_bounds__(256) void k_apply3_cf_fields(const int32_t *__restrict__ idx, const double *__restrict__ w, int nx, int ny, int ntx,
                                                          double scale, int sbe, int dbe, FieldTab tab) {
  constexpr int RPT = 2, TY = 4 * RPT;
  const Swz zs = make_swz(sbe), zd = make_swz(dbe);
  const int64_t P = (int64_t)nx * ny;
  const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = tile % ntx, ty = tile / ntx;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j0 = ty * TY + wave * RPT;
  int32_t c[RPT][3];
  double ww[RPT][3];
  bool act[RPT], mapped[RPT];
  int64_t po[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int j = j0 + r;
    const int i = tx * 64 + lane - mpg_tile_shift(j, nx);
    act[r] = (i >= 0) && (i < nx) && (j < ny);
    const int64_t p = act[r] ? (int64_t)j * nx + i : 0;
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
  for (int f = 0; f < tab.n; ++f) {
    const TS *s = (const TS *)tab.src[f];
    TD *d = (TD *)tab.dst[f];
    const double offset = tab.off[f];
    __syncthreads();
    double v[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const double a = (double)swz<SWZ>(s[c[r][0]], zs), b = (double)swz<SWZ>(s[c[r][1]], zs), e = (double)swz<SWZ>(s[c[r][2]], zs);
      v[r] = wsum3(ww[r][0], a, ww[r][1], b, ww[r][2], e);
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r)
      if (act[r]) stream_store_lane(swz<SWZ>((TD)fma(mapped[r] ? v[r] : 0.0, scale, offset), zd), d + po[r], (unsigned)lane * (unsigned)sizeof(TD));   // geom.h: per lane
  }
}

template <typename TS, typename TD, bool SWZ>
__global__ __launch_bounds__(256) void k_apply3_lf_t(const int32_t *__restrict__ idx, const double *__restrict__ w,
                                                     const TS *__restrict__ src, TD *__restrict__ dst, int nx, int ny, int64_t nsrc,
                                                     int nlev, int ntx, int nty, double scale, double offset, int sbe, int dbe, FieldTab tab) {
  extern __shared__ double sw[];    // sw[3][64] | sidx[3][64] | tile[nlev][65] in the DESTINATION type (narrowing at the tile
  int32_t *sidx = (int32_t *)(sw + 192);            // write or at the store gives the same bits; float32 halves the LDS -> 8 WGs / CU)
  TD *tile = (TD *)(sidx + 192);
  const Swz zs = make_swz(sbe), zd = make_swz(dbe);
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
  const TS *sf = mpg_field_src(tab, src, fld, (int64_t)nlev * nsrc);
  offset = mpg_field_off(tab, fld, offset);
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
        double a = (double)swz<SWZ>(r0[kk], zs), b = (double)swz<SWZ>(r1[kk], zs), e = (double)swz<SWZ>(r2[kk], zs);
        v[u] = fma(m ? wsum3(w0, a, w1, b, w2, e) : 0.0, scale, offset);
      }
      if (kact) {
#pragma unroll
        for (int u = 0; u < 4; ++u) tile[k * 65 + wave * 16 + q0 + u] = swz<SWZ>((TD)v[u], zd);
      }
    }
  }
  __syncthreads();
  TD *df = mpg_field_dst(tab, dst, fld, (int64_t)nlev * P);
  int j = ty, i = tx * 64 + lane - mpg_tile_shift(j, nx);
  if (i >= 0 && i < nx && j < ny) {
    int64_t p = (int64_t)j * nx + i;
    for (int k = wave; k < nlev; k += 4) stream_store_lane(tile[k * 65 + lane], df + (int64_t)k * P + p, (unsigned)lane * (unsigned)sizeof(TD));   // geom.h: per lane
  }
}

// The level-fast row gather on LINEAR tiles.  A tile is 64 consecutive target points of the flattened [ny][nx] plane
// starting at a multiple of 64, so every store of a level is one 256-byte (float32) / 512-byte (float64) segment that is
// naturally aligned whatever nx is (64 x 1 tiles of a 1800-wide grid start 32 bytes off a line in three rows of four:
// 10 % more bytes written, PMC, and two partial lines per store) -- IF the level's plane starts on a line, which only
// planes of a multiple of 32 (16) points do: the row block of a sharded job (133 x 1800), a stagger (1801 x 1060) or any
// odd grid puts level k's plane k * P * sizeof(TD) mod 128 bytes into one, and non-temporal stores of such segments cost 14-29 %
// of the kernel (profiles/r06_plane_alignment.md).  geom.h decides: float32 results per level, float64 results per lane.  Phase 0 stages the tile's three cell offsets
// (premultiplied by nlev, 32 bit, added to a scalar field base: no 64-bit address arithmetic per load) and weights in
// LDS; phase 1: a half-wave covers 64 levels of one point with ONE load per row, two levels per lane (8 bytes of a
// float32 row, 16 of a float64 row; rows are only element-aligned -- 55 float32 levels are 220 bytes -- which the
// hardware's unaligned access mode takes; the lane at the end of a row reads its last two levels and shifts), two points
// per wavefront pass, results transposed through an LDS tile [nlev][65] in the DESTINATION type (8 workgroups per CU
// with registers capped at 64); phase 2: lanes = points, non-temporal stores.  EPI = false leaves the affine epilogue
// out (mpg_regrid_dev: the float64 result as it stands, sign of zero included).  Round 2 measurements behind the shape
// (unroll, occupancy, fields per workgroup, 16-byte loads): profiles/r02_f32_row_gather.txt.
typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef double f64x2_u __attribute__((ext_vector_type(2), aligned(8)));
template <typename TS> struct Row2;
template <> struct Row2<float> { typedef f32x2_u type; };
template <> struct Row2<double> { typedef f64x2_u type; };

template <typename TS, typename TD, int UNR, bool EPI, bool SWZ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_apply3_lf_rows(
    const int32_t *__restrict__ idx, const double *__restrict__ w, const TS *__restrict__ src, TD *__restrict__ dst, int64_t P, int64_t nsrc,
    int nlev, unsigned ntile, double scale, double offset, int sbe, int dbe, int band, int st_mode, FieldTab tab) {
  typedef typename Row2<TS>::type row2;
  extern __shared__ double sw[];                    // sw[3][64] | soff[3][64] | tile[nlev][65] in the destination type
  uint32_t *soff = (uint32_t *)(sw + 192);
  TD *tile = (TD *)(soff + 192);
  const Swz zs = make_swz(sbe), zd = make_swz(dbe);
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  unsigned tl;
  int f;
  band_map(lin, ntile, gridDim.x / ntile, (unsigned)band, tl, f);
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
  const TS *sf = mpg_field_src(tab, src, f, (int64_t)nlev * nsrc);
  if constexpr (EPI) offset = mpg_field_off(tab, f, offset);
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
        const double y0 = (double)swz<SWZ>((TS)x[u][0].y, zs), y1 = (double)swz<SWZ>((TS)x[u][1].y, zs), y2 = (double)swz<SWZ>((TS)x[u][2].y, zs);
        const double a = shifted ? y0 : (double)swz<SWZ>((TS)x[u][0].x, zs), b = shifted ? y1 : (double)swz<SWZ>((TS)x[u][1].x, zs),
                     e = shifted ? y2 : (double)swz<SWZ>((TS)x[u][2].x, zs);
        const double r0 = wsum3(ww[u][0], a, ww[u][1], b, ww[u][2], e);
        const double r1 = wsum3(ww[u][0], y0, ww[u][1], y1, ww[u][2], y2);
        double v0 = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(r0) & keep[u]));
        double v1 = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(r1) & keep[u]));
        if constexpr (EPI) {
          v0 = fma(v0, scale, offset);
          v1 = fma(v1, scale, offset);
        }
        tile[kw0 + pt] = swz<SWZ>((TD)v0, zd);
        tile[kw1 + pt] = swz<SWZ>((TD)v1, zd);
      }
    }
  }
  __syncthreads();
  TD *df = mpg_field_dst(tab, dst, f, (int64_t)nlev * P) + p0;
  if (p0 + lane < P)
    for (int k = wave; k < nlev; k += 4) {
      TD *row = df + (int64_t)k * P;   // wave-uniform.  st_mode: the "lf_rows_store" knob -- 0: float64 results per lane (+22 % on planes 8 / 24 bytes off a line), float32 per level
      // (their 256-byte runs hold one whole line at most: per lane and per level measured equal); 1 plain, 2 non-temporal, 3 per lane (A/B)
      const bool per_lane = st_mode == 3 || (st_mode == 0 && sizeof(TD) == 8);
      stream_store(tile[k * 65 + lane], row + lane, per_lane ? stream_lane_full(row + lane, (unsigned)lane * (unsigned)sizeof(TD)) : st_mode ? st_mode == 2 : stream_nt(row));
    }
}

// nearest (NNZ = 1, weights implicit), 4-point destagger (NNZ = 4) and CSR (NNZ = 0): one thread per target point.  The
// fixed forms keep the point's indices and weights in registers for all levels and work on two levels at a time (2 x NNZ
// independent loads in flight); accumulation order per level as k_applyN / k_apply1 of k_apply.hip -> the same bits.
template <typename TS, typename TD, bool SWZ, int NNZ, bool LEVF>
__global__ __launch_bounds__(256) void k_apply_generic_t(const int32_t *__restrict__ idx, const double *__restrict__ w,
                                                         const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                         const double *__restrict__ val, const TS *__restrict__ src,
                                                         TD *__restrict__ dst, int64_t P, int64_t nsrc, int nlev, int nblk,
                                                         double scale, double offset, int sbe, int dbe, FieldTab tab) {
  constexpr bool lev_fast = LEVF;   // the layout is part of the instantiation: no per-level branch on it (round-5 review, item 6)
  const Swz zs = make_swz(sbe), zd = make_swz(dbe);
  unsigned blk = blockIdx.x % nblk;
  int fld = blockIdx.x / nblk;
  int64_t p = (int64_t)blk * 256 + threadIdx.x;
  if (p >= P) return;
  const TS *sf = mpg_field_src(tab, src, fld, (int64_t)nlev * nsrc);
  TD *df = mpg_field_dst(tab, dst, fld, (int64_t)nlev * P);
  offset = mpg_field_off(tab, fld, offset);
  if constexpr (NNZ == 0) {
    for (int k = 0; k < nlev; ++k) {
      double acc = 0.0;
      for (int q = rowptr[p]; q < rowptr[p + 1]; ++q) {
        int32_t c = col[q];
        acc = fma(val[q], (double)swz<SWZ>(lev_fast ? sf[(int64_t)c * nlev + k] : sf[(int64_t)k * nsrc + c], zs), acc);
      }
      stream_store_lane(swz<SWZ>((TD)fma(acc, scale, offset), zd), df + (int64_t)k * P + p, (unsigned)(threadIdx.x & 63) * (unsigned)sizeof(TD));
    }
  } else {
    int32_t c[NNZ];
    double ww[NNZ];
#pragma unroll
    for (int q = 0; q < NNZ; ++q) {
      c[q] = idx[q * P + p];
      ww[q] = NNZ == 1 ? 1.0 : w[q * P + p];
    }
    const bool mapped = c[0] >= 0;
    int64_t step = lev_fast ? 1 : nsrc, base[NNZ];
#pragma unroll
    for (int q = 0; q < NNZ; ++q) base[q] = mapped ? (lev_fast ? (int64_t)c[q] * nlev : (int64_t)c[q]) : 0;
    auto level = [&](int k, double *v) {
#pragma unroll
      for (int q = 0; q < NNZ; ++q) v[q] = (double)swz<SWZ>(sf[base[q] + k * step], zs);
    };
    auto combine = [&](const double *v) -> double {
      double acc = 0.0;
      if (NNZ == 1) acc = v[0];
      else {
#pragma unroll
        for (int q = 0; q < NNZ; ++q) acc = fma(ww[q], v[q], acc);
      }
      return mapped ? acc : 0.0;
    };
    // Two levels per step, and the loads of the NEXT two are issued before the stores of the current two: loads and stores
    // share one in-order counter on gfx950, so a load issued after a store cannot be consumed before that store has been
    // acknowledged -- with the loads in front, a step waits for its own data only (s_waitcnt vmcnt(2): the two stores stay
    // in flight).
    auto put = [&](int k, const double *v) { stream_store_lane(swz<SWZ>((TD)fma(combine(v), scale, offset), zd), df + (int64_t)k * P + p, (unsigned)(threadIdx.x & 63) * (unsigned)sizeof(TD)); };
    double a0[NNZ], a1[NNZ];
    level(0, a0);
    level(nlev > 1 ? 1 : 0, a1);
    int k = 0;
    for (; k + 3 < nlev; k += 2) {
      double b0[NNZ], b1[NNZ];
      level(k + 2, b0);
      level(k + 3, b1);
      put(k, a0);
      put(k + 1, a1);
#pragma unroll
      for (int q = 0; q < NNZ; ++q) {
        a0[q] = b0[q];
        a1[q] = b1[q];
      }
    }
    // a0 / a1 hold levels k, k + 1; at most one more level (k + 2) is left
    if (k + 2 < nlev) {
      double b0[NNZ];
      level(k + 2, b0);
      put(k, a0);
      put(k + 1, a1);
      put(k + 2, b0);
    } else {
      put(k, a0);
      if (k + 1 < nlev) put(k + 1, a1);
    }
  }
}

// the (field, tile) order of the row gather: bands of 1024 tiles (64 K points, 2.4 MB of indices + weights: they stay in the
// XCD's L2 from one field of the bundle to the next), all fields of a band before the next band -- 4 % on configuration 4
#define LF_ROWS_BAND 1024
static int g_lf_rows_store = 0;   // "lf_rows_store" knob: 0 = float32 results per level by the alignment of its plane (geom.h stream_nt), float64 per lane (stream_lane_full); 1 = plain, 2 = non-temporal, 3 = per lane (A/B)
void mpg_set_lf_rows_store(int v) { g_lf_rows_store = v; }
static size_t lf_rows_lds(size_t dst_size, int nlev) { return dst_size * 65 * (size_t)nlev + sizeof(double) * 192 + sizeof(int32_t) * 192; }
static bool lf_rows_fits(const mpg_handle_s *h, size_t dst_size, int nlev) {
  return nlev >= 2 && (uint64_t)h->n_src * (uint64_t)nlev < 0xFFFFFFFFull && lf_rows_lds(dst_size, nlev) <= 160 * 1024;
}

template <typename TS, typename TD, bool SWZ>
static int launch_typed(mpg_handle_s *h, const void *src, int layout, int nlev, int nfields, void *dst, double scale, double offset, int sbe,
                        int dbe, hipStream_t s, const FieldTab &tab) {
  int64_t P = h->n_dst;
  int lev_fast = layout == MPG_LAYOUT_LEV_FAST && nlev > 1;
  if (h->kind == MPG_KIND_FIXED && h->nnz_per_row == 3) {
    if (lev_fast && mpg_lf_variant() != MPG_LF_ROWTILES && lf_rows_fits(h, sizeof(TD), nlev)) {
      const size_t lds = std::min<size_t>(lf_rows_lds(sizeof(TD), nlev) + (size_t)mpg_staged_lds_pad_kb() * 1024, 160 * 1024);   // the pad: A/B knob, 0 in production
      const unsigned ntile = (unsigned)((P + 63) / 64);
      auto fn = k_apply3_lf_rows<TS, TD, sizeof(TS) == 4 ? 2 : 1, true, SWZ>;   // measured: unroll 2 for float32 rows, 1 for float64
      if (lds > 48 * 1024) MPG_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      fn<<<ntile * (unsigned)nfields, 256, lds, s>>>(h->idx.p, h->w.p, (const TS *)src, (TD *)dst, P, h->n_src, nlev, ntile, scale, offset, sbe, dbe, mpg_field_band(LF_ROWS_BAND), g_lf_rows_store, tab);
    } else if (lev_fast) {
      size_t lds = sizeof(TD) * 65 * (size_t)nlev + sizeof(double) * 192 + sizeof(int32_t) * 192;
      if (lds > 160 * 1024) {
        mpg_set_error("Regrid(LEV_FAST): %d levels exceed the LDS tile", nlev);
        return MPG_ERR_UNSUPPORTED;
      }
      auto fn = k_apply3_lf_t<TS, TD, SWZ>;
      if (lds > 48 * 1024) MPG_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const int ntxs = mpg_tile_ntx(h->nx_dst, 64), nty = h->ny_dst;
      fn<<<(unsigned)ntxs * nty * nfields, 256, lds, s>>>(h->idx.p, h->w.p, (const TS *)src, (TD *)dst, h->nx_dst, h->ny_dst, h->n_src, nlev, ntxs, nty,
                                                         scale, offset, sbe, dbe, tab);
    } else {
      int ntx = mpg_tile_ntx(h->nx_dst, 64), nty = (h->ny_dst + 7) / 8;
      k_apply3_cf_t<TS, TD, SWZ><<<(unsigned)ntx * nty * nfields, 256, 0, s>>>(h->idx.p, h->w.p, (const TS *)src, (TD *)dst, h->nx_dst, h->ny_dst,
                                                                              h->n_src, nlev, ntx, nty, scale, offset, sbe, dbe, tab);
    }
  } else {
    int nblk = (int)((P + 255) / 256);
    const int nnz = h->kind == MPG_KIND_CSR ? 0 : h->nnz_per_row;
    if (nnz != 0 && nnz != 1 && nnz != 4) {
      mpg_set_error("Regrid: unsupported handle (%d weights per row)", nnz);
      return MPG_ERR_UNSUPPORTED;
    }
    auto fn = lev_fast ? (nnz == 0 ? k_apply_generic_t<TS, TD, SWZ, 0, true> : (nnz == 1 ? k_apply_generic_t<TS, TD, SWZ, 1, true> : k_apply_generic_t<TS, TD, SWZ, 4, true>))
                       : (nnz == 0 ? k_apply_generic_t<TS, TD, SWZ, 0, false> : (nnz == 1 ? k_apply_generic_t<TS, TD, SWZ, 1, false> : k_apply_generic_t<TS, TD, SWZ, 4, false>));
    fn<<<(unsigned)nblk * nfields, 256, 0, s>>>(h->idx.p, h->w.p, h->rowptr.p, h->col.p, h->val.p, (const TS *)src, (TD *)dst, P, h->n_src, nlev,
                                               nblk, scale, offset, sbe, dbe, tab);
  }
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

// float64 rows in, float64 out, no epilogue: the level-fast row gather of mpg_regrid_dev (k_apply.hip).
// -> MPG_ERR_UNSUPPORTED when the 32-bit row offsets or the LDS tile do not fit (the caller keeps its older kernel).
int mpg_k_apply3_lf_rows(mpg_handle_s *h, const double *src, int nlev, int nfields, double *dst, hipStream_t s) {
  const int64_t P = h->n_dst;
  if (!lf_rows_fits(h, sizeof(double), nlev)) return MPG_ERR_UNSUPPORTED;
  const size_t lds = lf_rows_lds(sizeof(double), nlev);
  const unsigned ntile = (unsigned)((P + 63) / 64);
  auto fn = k_apply3_lf_rows<double, double, 1, false, false>;
  if (lds > 48 * 1024) MPG_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  fn<<<ntile * (unsigned)nfields, 256, lds, s>>>(h->idx.p, h->w.p, src, dst, P, h->n_src, nlev, ntile, 1.0, 0.0, 0, 0, mpg_field_band(LF_ROWS_BAND), g_lf_rows_store, FieldTab());
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

template <bool SWZ>
static int launch_typed_types(mpg_handle_s *h, const void *src, int sf32, int layout, int nlev, int nfields, void *dst, int df32, double scale,
                              double offset, int sbe, int dbe, hipStream_t s, const FieldTab &tab) {
  if (sf32 && df32) return launch_typed<float, float, SWZ>(h, src, layout, nlev, nfields, dst, scale, offset, sbe, dbe, s, tab);
  if (sf32) return launch_typed<float, double, SWZ>(h, src, layout, nlev, nfields, dst, scale, offset, sbe, dbe, s, tab);
  if (df32) return launch_typed<double, float, SWZ>(h, src, layout, nlev, nfields, dst, scale, offset, sbe, dbe, s, tab);
  return launch_typed<double, double, SWZ>(h, src, layout, nlev, nfields, dst, scale, offset, sbe, dbe, s, tab);
}

int mpg_k_apply_typed(mpg_handle_s *h, const void *src, int src_type, int layout, int nlev, int nfields, void *dst, int dst_type,
                      double scale, double offset, hipStream_t s, const FieldTab &tab) {
  if (h->n_dst == 0 || nlev == 0 || nfields == 0) return MPG_SUCCESS;
  const int sf32 = src_type & MPG_TYPE_F32, df32 = dst_type & MPG_TYPE_F32;
  const int sbe = (src_type & MPG_TYPE_BE) != 0, dbe = (dst_type & MPG_TYPE_BE) != 0;
  if (h->n_src == 0) {  // nothing mapped: the destination is the epilogue of 0.0
    if (offset != 0.0) {
      mpg_set_error("mpg_regrid_typed: handle without sources and a non-zero offset is not supported");
      return MPG_ERR_UNSUPPORTED;
    }
    if (tab.n) return MPG_ERR_UNSUPPORTED;   // (mpg_regrid_bundle_typed_dev serves such a handle field by field)
    MPG_HIP(hipMemsetAsync(dst, 0, (df32 ? 4 : 8) * (size_t)h->n_dst * nlev * nfields, s));
    return MPG_SUCCESS;
  }
  const bool three = h->kind == MPG_KIND_FIXED && h->nnz_per_row == 3;
  const bool lev_fast = layout == MPG_LAYOUT_LEV_FAST && nlev > 1;
  const bool long_bundle = nlev * nfields >= MPG_STAGE_MIN_LEVELS;
  if (three && nlev == 1 && tab.n > 1) {   // 2-D fields in separate arrays: the bundle's fields are walked like levels
    const int ntx = mpg_tile_ntx(h->nx_dst, 64), nty = (h->ny_dst + 7) / 8;
    const unsigned nwg = (unsigned)ntx * nty;
#define CF_FIELDS(TS, TD)                                                                                                                                        \
  do {                                                                                                                                                           \
    if (sbe || dbe) k_apply3_cf_fields<TS, TD, true><<<nwg, 256, 0, s>>>(h->idx.p, h->w.p, h->nx_dst, h->ny_dst, ntx, scale, sbe, dbe, tab);                      \
    else k_apply3_cf_fields<TS, TD, false><<<nwg, 256, 0, s>>>(h->idx.p, h->w.p, h->nx_dst, h->ny_dst, ntx, scale, 0, 0, tab);                                   \
  } while (0)
    if (sf32 && df32) CF_FIELDS(float, float);
    else if (sf32) CF_FIELDS(float, double);
    else if (df32) CF_FIELDS(double, float);
    else CF_FIELDS(double, double);
#undef CF_FIELDS
    MPG_HIP(hipGetLastError());
    return MPG_SUCCESS;
  }
  if (three && !lev_fast && !sbe && !dbe) {   // cell-fast: LDS-staged (k_apply_lfu.hip) unless the knob or the handle says lane gather
    int staged = mpg_a3_staged(), rc;
    if (staged == -1) {
      staged = -2;
      if (long_bundle || h->cf_choice > 0) {
        if ((rc = mpg_cfu_auto(h, s, &staged))) return rc;
      }
    } else if (staged >= 0) {
      int fits;
      if ((rc = mpg_cfu_fits(h, staged, s, &fits))) return rc;
      if (!fits) staged = -2;
    }
    if (staged >= 0) {
      rc = mpg_k_apply3_cfu(h, staged, src, sf32, nlev, nfields, dst, df32, true, scale, offset, s, tab);
      if (rc != MPG_ERR_UNSUPPORTED) return rc;
    }
  }
  if (three && lev_fast) {
    int lfv = mpg_lf_variant(), rc;
    if (lfv < 0) {
      lfv = MPG_LF_ROWS;
      if (long_bundle && (rc = mpg_lfu_auto(h, s, &lfv))) return rc;
    }
    if (lfv == MPG_LF_STAGED) {
      rc = mpg_k_apply3_lfu_typed(h, src, src_type, nlev, nfields, dst, dst_type, scale, offset, s, tab);
      if (rc != MPG_ERR_UNSUPPORTED) return rc;
    }
  }
  int rc = (sbe || dbe) ? launch_typed_types<true>(h, src, sf32, layout, nlev, nfields, dst, df32, scale, offset, sbe, dbe, s, tab)
                        : launch_typed_types<false>(h, src, sf32, layout, nlev, nfields, dst, df32, scale, offset, 0, 0, s, tab);
  if (rc == MPG_SUCCESS && h->n_pole) rc = mpg_k_pole_fix(h, src, src_type, layout, nlev, nfields, dst, dst_type, scale, offset, s, tab);
  return rc;
}

// mpg_init loads this translation unit's code object ahead of its first launch (mpg_api.hip: warm_modules)
const void *mpg_anchor_k_apply_typed() { return (const void *)&k_apply3_cf_t<float, float, false>; }
