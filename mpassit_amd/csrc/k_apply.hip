// Weight application ("Regrid") kernels -- the HBM-bound hot path.
//
// Replace ESMF_Field[Bundle]Regrid at interp.F90:134,219,236,251,268,286,307,325,344,363,382,404,431,
// 443.  Semantics: dst(p,k) = sum_j w_pj * src(c_pj, k) for every level k of the
// ungridded dimension, float64 accumulation; the destination is fully overwritten and unmapped points
// are 0.0 (zeroregion=TOTAL + unmappedaction=IGNORE); nearest-neighbour is a pure copy (bit exact).
//
//   K2  k_apply3_cf    3-point gather, source cell-fastest [nlev][ncell] (reference memory order,
//                       input_data.F90:653-655), destination [nlev][ny][nx]: the lane-gather form; the default
//                       for cell-fast bundles is the LDS-staged k_apply3_cfu of k_apply_lfu.hip (chosen per
//                       handle in mpg_k_apply below), this one serves short bundles (2-D fields) and handles
//                       whose tiles share no cells
//   K2' k_apply3_lf    same from level-fastest [ncell][nlev] (MPAS file order, input_data.F90:630,645):
//                       the reference's host transpose is fused away through an LDS tile transpose; the older
//                       row-gather form on grid-row tiles (default: k_apply3_lf_rows, k_apply_typed.hip, when
//                       target points share few cells, else k_apply3_lfu)
//   K3  k_apply1       nearest-neighbour copy
//   K4  k_apply_csr    conservative (variable row length)
//   K6  k_applyN<4>    4-point destagger (CENTER -> EDGE1/EDGE2)
//   K7  k_rotate       rotate_winds_cgrid (interp.F90:737-748)
//
// Roofline: no reuse beyond the ~1.9 target points that share a source value, 5 flop per 36-60 B ->
// HBM-bound; MFMA does not apply.  Design for CDNA4: 64 consecutive i per wave (512 B coalesced,
// non-temporal stores so the write stream does not evict the source lines from L2), 2-D target tiles so
// a workgroup's gather footprint is spatially compact, XCD-aware tile order so neighbouring tiles share
// an L2, weights/indices SoA and read once per tile for all levels.
#include <string.h>

#include "geom.h"
#include "mpg_internal.h"

#define A3_TX 64

// The lane-gather form: a 256-thread workgroup = 64 (i) x 8 (j) target points, two rows per thread; the waves are kept
// in level lock-step by one barrier per level so that lines shared between neighbouring rows are still in L1/L2 when the
// next wave asks for them.  Indices and weights are read once per tile and kept in registers for all levels; destination
// stores are non-temporal.  (Shapes measured in round 1 and dropped from the library in round 3 -- more rows per thread,
// 8 / 16 waves, level chunks, several fields per workgroup, banded tile order, 16 x 4 / 32 x 2 wave patches: all equal or
// slower, profiles/r01_sweep_apply*.txt.)
__global__ __launch_bounds__(256) void k_apply3_cf(const int32_t *__restrict__ idx, const double *__restrict__ w,
                                                   const double *__restrict__ src, double *__restrict__ dst, int nx, int ny,
                                                   int64_t nsrc, int nlev, int ntx, int nty) {
  constexpr int RPT = 2, TY = 4 * RPT;
  int64_t P = (int64_t)nx * ny;
  unsigned ntile = (unsigned)ntx * nty;
  unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  unsigned tile = lin % ntile;
  int f = lin / ntile;
  int tx = tile % ntx, ty = tile / ntx;
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int ib = tx * A3_TX + lane;
  int j0 = ty * TY + wave * RPT;

  int32_t c[RPT][3];
  double ww[RPT][3];
  bool act[RPT], mapped[RPT];
  int64_t po[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    int j = j0 + r;
    int i = ib - mpg_tile_shift(j, nx);       // row-shifted tile: aligned store segments (mpg_internal.h)
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
  const double *s = src + (int64_t)f * nlev * nsrc;
  double *d = dst + (int64_t)f * nlev * P;
  for (int k = 0; k < nlev; ++k) {
    __syncthreads();
    double v[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      double a = s[c[r][0]], b = s[c[r][1]], e = s[c[r][2]];
      v[r] = wsum3(ww[r][0], a, ww[r][1], b, ww[r][2], e);
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r)
      if (act[r]) stream_store_lane(mapped[r] ? v[r] : 0.0, d + po[r], (unsigned)(threadIdx.x & 63) * 8u);   // geom.h: per lane
    s += nsrc;
    d += P;
  }
}
static int g_a3_staged = -1;  // "a3_staged" knob: -2 lane-gather only, -1 per-handle choice (default), 0..2 that LDS-staged variant

// Level-fastest source ([ncell][nlev], MPAS file order) on tiles of 64 points of ONE grid row: the older row gather, kept as
// the route for handles the default (k_apply3_lf_rows, k_apply_typed.hip: linear tiles, 32-bit row offsets) cannot take --
// n_src * nlev >= 2^32 or a single level -- and as its cross-check ("lf_variant" 2).
// phase 0: the tile's 64 x 3 indices/weights are staged in LDS (coalesced);
// phase 1: wave w serves points 8w..8w+7, lanes = levels: the cell id is wave-uniform (readfirstlane -> scalar row base),
//          so each gather is one coalesced nlev*8-byte row read; 4 points (12 loads) in flight;
// phase 2: lanes = points: 512-byte contiguous non-temporal stores per level.
// LDS tile [nlev][65] doubles (row pad 1: conflict-free ds_write_b64 column writes).
__global__ __launch_bounds__(512) void k_apply3_lf(const int32_t *__restrict__ idx, const double *__restrict__ w,
                                                   const double *__restrict__ src, double *__restrict__ dst, int nx, int ny,
                                                   int64_t nsrc, int nlev, int ntx, int nty) {
  constexpr int WAVES = 8, PPW = 64 / WAVES, BATCH = 4;
  extern __shared__ double tile[];  // [nlev][65] | sw[3][64] | sidx[3][64]
  double *sw = tile + (size_t)nlev * 65;
  int32_t *sidx = (int32_t *)(sw + 192);
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
  int oj = ty, oi = tx * 64 + lane - mpg_tile_shift(oj, nx);
  bool oact = oi >= 0 && oi < nx && oj < ny;
  int64_t op = oact ? (int64_t)oj * nx + oi : 0;
  const double *sf = src + (int64_t)fld * nlev * nsrc;
  for (int kb = 0; kb < nlev; kb += 64) {
    int k = kb + lane;
    bool kact = k < nlev;
    int kk = kact ? k : 0;
#pragma unroll
    for (int q0 = 0; q0 < PPW; q0 += BATCH) {
      double v[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        int pt = wave * PPW + q0 + u;
        int32_t c0 = __builtin_amdgcn_readfirstlane(sidx[pt]);
        int32_t c1 = __builtin_amdgcn_readfirstlane(sidx[64 + pt]);
        int32_t c2 = __builtin_amdgcn_readfirstlane(sidx[128 + pt]);
        double w0 = sw[pt], w1 = sw[64 + pt], w2 = sw[128 + pt];
        bool m = c0 >= 0;
        c0 = max(c0, 0); c1 = max(c1, 0); c2 = max(c2, 0);
        const double *r0 = sf + (int64_t)c0 * nlev, *r1 = sf + (int64_t)c1 * nlev, *r2 = sf + (int64_t)c2 * nlev;
        double a = r0[kk], b = r1[kk], e = r2[kk];
        v[u] = m ? wsum3(w0, a, w1, b, w2, e) : 0.0;
      }
      if (kact) {
#pragma unroll
        for (int u = 0; u < BATCH; ++u) tile[k * 65 + wave * PPW + q0 + u] = v[u];
      }
    }
  }
  __syncthreads();
  double *df = dst + (int64_t)fld * nlev * P;
  if (oact)
    for (int k = wave; k < nlev; k += WAVES) stream_store_lane(tile[k * 65 + lane], df + (int64_t)k * P + op, (unsigned)lane * 8u);
}
// "lf_variant" knob: -1 per-handle choice (default) between 0 and 1; 0 row gather on linear aligned tiles
// (k_apply3_lf_rows), 1 level-chunked LDS-staged kernel (k_apply_lfu.hip), 2 row gather on grid-row tiles (k_apply3_lf /
// k_apply3_lf_t: the capacity fallback)
static int g_lf_variant = -1;

// nearest neighbour (bit-exact copy), 4-point destagger and conservative CSR: one thread per target point.  LEVF (the source is in file
// order, [cell][lev]) is a template parameter -- no per-level branch on the layout -- and the stores are non-temporal: a result is
// written once and never read by this kernel, it must not push the gathered source lines out of L2 (round-5 review, item 6).
template <bool LEVF>
__global__ __launch_bounds__(256) void k_apply1(const int32_t *__restrict__ idx, const double *__restrict__ src,
                                                double *__restrict__ dst, int64_t P, int64_t nsrc, int nlev, int nblk) {
  unsigned blk = blockIdx.x % nblk;
  int fld = blockIdx.x / nblk;
  int64_t p = (int64_t)blk * 256 + threadIdx.x;
  if (p >= P) return;
  int32_t c = idx[p];
  const double *sf = src + (int64_t)fld * nlev * nsrc + (c >= 0 ? (LEVF ? (int64_t)c * nlev : (int64_t)c) : 0);
  const int64_t step = LEVF ? 1 : nsrc;
  double *df = dst + (int64_t)fld * nlev * P + p;
  for (int k = 0; k < nlev; ++k) {
    double v = 0.0;
    if (c >= 0) v = sf[k * step];
    stream_store_lane(v, df + (int64_t)k * P, (unsigned)(threadIdx.x & 63) * 8u);
  }
}

template <int NNZ, bool LEVF>
__global__ __launch_bounds__(256) void k_applyN(const int32_t *__restrict__ idx, const double *__restrict__ w,
                                                const double *__restrict__ src, double *__restrict__ dst, int64_t P,
                                                int64_t nsrc, int nlev, int nblk) {
  unsigned blk = blockIdx.x % nblk;
  int fld = blockIdx.x / nblk;
  int64_t p = (int64_t)blk * 256 + threadIdx.x;
  if (p >= P) return;
  int64_t c[NNZ];
  double ww[NNZ];
#pragma unroll
  for (int q = 0; q < NNZ; ++q) {
    const int32_t ci = idx[q * P + p];
    c[q] = ci < 0 ? -1 : (LEVF ? (int64_t)ci * nlev : (int64_t)ci);
    ww[q] = w[q * P + p];
  }
  bool mapped = c[0] >= 0;
  const int64_t step = LEVF ? 1 : nsrc;
  const double *sf = src + (int64_t)fld * nlev * nsrc;
  double *df = dst + (int64_t)fld * nlev * P + p;
  for (int k = 0; k < nlev; ++k) {
    double acc = 0.0;
    if (mapped) {
#pragma unroll
      for (int q = 0; q < NNZ; ++q) acc = fma(ww[q], sf[c[q] + k * step], acc);
    }
    stream_store_lane(acc, df + (int64_t)k * P, (unsigned)(threadIdx.x & 63) * 8u);
  }
}

template <bool LEVF>
__global__ __launch_bounds__(256) void k_apply_csr(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                   const double *__restrict__ val, const double *__restrict__ src,
                                                   double *__restrict__ dst, int64_t P, int64_t nsrc, int nlev, int nblk) {
  unsigned blk = blockIdx.x % nblk;
  int fld = blockIdx.x / nblk;
  int64_t p = (int64_t)blk * 256 + threadIdx.x;
  if (p >= P) return;
  int b = rowptr[p], e = rowptr[p + 1];
  const double *sf = src + (int64_t)fld * nlev * nsrc;
  double *df = dst + (int64_t)fld * nlev * P + p;
  for (int k = 0; k < nlev; ++k) {
    double acc = 0.0;
    for (int q = b; q < e; ++q) {
      int32_t c = col[q];
      acc = fma(val[q], LEVF ? sf[(int64_t)c * nlev + k] : sf[(int64_t)k * nsrc + c], acc);
    }
    stream_store_lane(acc, df + (int64_t)k * P, (unsigned)(threadIdx.x & 63) * 8u);
  }
}

// rotate_winds_cgrid (interp.F90:737-748); evaluated exactly as written (no FMA) -> bit-identical to the oracle
__global__ __launch_bounds__(256) void k_rotate(int64_t npts, int nlev, const double *__restrict__ cosa,
                                                const double *__restrict__ sina, double *__restrict__ u, double *__restrict__ v) {
#pragma clang fp contract(off)
  int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p >= npts) return;
  double ca = cosa[p], sa = sina[p];
  double tana = sa / ca;
  double den = ca + sa * tana;
  for (int k = 0; k < nlev; ++k) {
    int64_t q = (int64_t)k * npts + p;
    double uo = u[q], vo = v[q];
    double t1 = vo * tana;
    double un = (uo + t1) / den;
    double t2 = un * sa;
    double vn = (vo - t2) / ca;
    u[q] = un;
    v[q] = vn;
  }
}

__global__ __launch_bounds__(256) void k_pack(const double *__restrict__ src, int64_t nsrc, int nlev,
                                              const int32_t *__restrict__ ids, int64_t nids, double *__restrict__ dst) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nids) return;
  int32_t c = ids[i];
  for (int k = 0; k < nlev; ++k) dst[(int64_t)k * nids + i] = src[(int64_t)k * nsrc + c];
}

static int g_store_boxes = 1;
int mpg_store_boxes() { return g_store_boxes; }
static int g_bilinear_linetype = 0;
int mpg_bilinear_linetype() { return g_bilinear_linetype; }
static int g_node_fan_origin = 0;
int mpg_node_fan_origin() { return g_node_fan_origin; }
static int g_grid_inside_tol_exp = 10;
int mpg_grid_inside_tol_exp() { return g_grid_inside_tol_exp; }
int mpg_a3_staged() { return g_a3_staged; }
int mpg_lf_variant() { return g_lf_variant; }

int mpg_k_tune(const char *key, int value) {
  if (!strcmp(key, "lf_variant")) {
    if (value < -1 || value > 2) return MPG_ERR_INVALID_ARG;
    g_lf_variant = value;
    return MPG_SUCCESS;
  }
  if (!strcmp(key, "a3_staged")) {
    if (value < -2 || value >= mpg_cfu_num_variants()) return MPG_ERR_INVALID_ARG;
    g_a3_staged = value;
    return MPG_SUCCESS;
  }
  if (!strcmp(key, "bilinear_linetype")) {   // Mesh -> Grid bilinear Store: where the target point meets the triangle's plane
    if (value != 0 && value != 1) return MPG_ERR_INVALID_ARG;
    g_bilinear_linetype = value;
    return MPG_SUCCESS;
  }
  if (!strcmp(key, "node_fan_origin")) {   // node-located bilinear Store: which listed vertex of a polygon is the apex of its fan (-1: the last one)
    if (value < -8 || value > 15) return MPG_ERR_INVALID_ARG;
    g_node_fan_origin = value;
    return MPG_SUCCESS;
  }
  if (!strcmp(key, "grid_inside_tol_exp")) {   // Grid -> Grid Store: a stagger point is inside a quad of centres within 10^-value of its parametric range
    if (value < 3 || value > 16) return MPG_ERR_INVALID_ARG;
    g_grid_inside_tol_exp = value;
    return MPG_SUCCESS;
  }
  if (!strcmp(key, "store_boxes")) {   // Stores on projection-built grids: candidates from the inverse projection (1) or the pyramid walk (0)
    if (value != 0 && value != 1) return MPG_ERR_INVALID_ARG;
    g_store_boxes = value;
    return MPG_SUCCESS;
  }
  if (!strcmp(key, "nn_variant")) {   // nearest-neighbour Store: 1 = wave-cooperative search, 0 = one thread per point
    if (value != 0 && value != 1) return MPG_ERR_INVALID_ARG;
    mpg_set_nearest_variant(value);
    return MPG_SUCCESS;
  }
  if (!strcmp(key, "field_band")) {
    if (value < -1 || value > 65536) return MPG_ERR_INVALID_ARG;
    mpg_set_field_band(value);
    return MPG_SUCCESS;
  }
  if (!strcmp(key, "lfu_npf")) {   // row slots per thread of the staged level-fast kernel: 0 = by each tile's own list (round 6); 2 .. 32 = at least that many for every tile (A/B: 16 = rounds 1-4)
    if (value != 0 && value != 2 && value != 4 && value != 8 && value != 16 && value != 32) return MPG_ERR_INVALID_ARG;
    mpg_lfu_set_npf(value);
    return MPG_SUCCESS;
  }
  if (!strcmp(key, "staged_lds_pad_kb")) {   // A/B only: extra dynamic LDS (KB) for the staged Regrid kernels = fewer workgroups per CU (the kernels live on L2 keeping what
    if (value < -1 || value > 128) return MPG_ERR_INVALID_ARG;   // they stream, profiles/r06_src_nt_loads.txt: does a smaller in-flight working set pay for the latency hiding it costs?)
    mpg_set_staged_lds_pad_kb(value < 0 ? 0 : value);
    return MPG_SUCCESS;
  }
  if (!strcmp(key, "staged_store")) {   // A/B only: stores of the staged cell-fast kernel: 0 = per lane (geom.h stream_store_lane), 2 = every lane non-temporal
    if (value != 0 && value != 2) return MPG_ERR_INVALID_ARG;
    mpg_set_staged_store(value);
    return MPG_SUCCESS;
  }
  if (!strcmp(key, "lf_rows_store")) {   // store policy of the level-fast row gather: 0 = float32 results per level by the alignment of its plane, float64 per lane (geom.h); 1 = plain, 2 = non-temporal, 3 = per lane (A/B)
    if (value < 0 || value > 3) return MPG_ERR_INVALID_ARG;
    mpg_set_lf_rows_store(value);
    return MPG_SUCCESS;
  }
  if (!strcmp(key, "lfu_min_reuse_x10")) {
    if (value < 0 || value > 1000) return MPG_ERR_INVALID_ARG;
    mpg_lfu_set_min_reuse_x10(value);
    return MPG_SUCCESS;
  }
  return MPG_ERR_INVALID_ARG;
}

int mpg_k_apply(mpg_handle_s *h, const double *src, int layout, int nlev, int nfields, double *dst, hipStream_t s) {
  int64_t P = h->n_dst;
  int lev_fast = layout == MPG_LAYOUT_LEV_FAST;
  if (P == 0 || nlev == 0 || nfields == 0) return MPG_SUCCESS;
  if (h->n_src == 0) {  // nothing is mapped (e.g. a row shard entirely outside the mesh): zero-filled destination
    MPG_HIP(hipMemsetAsync(dst, 0, sizeof(double) * (size_t)P * nlev * nfields, s));
    return MPG_SUCCESS;
  }
  int nblk = (int)((P + 255) / 256);
  const bool levf = lev_fast && nlev > 1;   // (a single level is the same memory in both layouts)
  if (h->kind == MPG_KIND_CSR) {
    if (levf) k_apply_csr<true><<<(unsigned)nblk * nfields, 256, 0, s>>>(h->rowptr.p, h->col.p, h->val.p, src, dst, P, h->n_src, nlev, nblk);
    else k_apply_csr<false><<<(unsigned)nblk * nfields, 256, 0, s>>>(h->rowptr.p, h->col.p, h->val.p, src, dst, P, h->n_src, nlev, nblk);
  } else if (h->nnz_per_row == 1) {
    if (levf) k_apply1<true><<<(unsigned)nblk * nfields, 256, 0, s>>>(h->idx.p, src, dst, P, h->n_src, nlev, nblk);
    else k_apply1<false><<<(unsigned)nblk * nfields, 256, 0, s>>>(h->idx.p, src, dst, P, h->n_src, nlev, nblk);
  } else if (h->nnz_per_row == 4) {
    if (levf) k_applyN<4, true><<<(unsigned)nblk * nfields, 256, 0, s>>>(h->idx.p, h->w.p, src, dst, P, h->n_src, nlev, nblk);
    else k_applyN<4, false><<<(unsigned)nblk * nfields, 256, 0, s>>>(h->idx.p, h->w.p, src, dst, P, h->n_src, nlev, nblk);
  } else if (h->nnz_per_row == 3 && lev_fast && nlev > 1) {
    int lfv = g_lf_variant;
    if (lfv < 0) {  // per handle, by the (sampled) reuse statistic of its tiles (k_apply_lfu.hip); short bundles: row gather
      lfv = MPG_LF_ROWS;
      if (nlev * nfields >= MPG_STAGE_MIN_LEVELS) {
        int rc = mpg_lfu_auto(h, s, &lfv);
        if (rc) return rc;
      }
    }
    if (lfv == MPG_LF_STAGED) {
      int rc = mpg_k_apply3_lfu(h, src, nlev, nfields, dst, s);
      if (rc != MPG_ERR_UNSUPPORTED) return rc;
      lfv = MPG_LF_ROWS;     // the tile lists outgrow the LDS: row gather
    }
    if (lfv == MPG_LF_ROWS) {
      int rc = mpg_k_apply3_lf_rows(h, src, nlev, nfields, dst, s);   // k_apply_typed.hip
      if (rc != MPG_ERR_UNSUPPORTED) return rc;
    }
    int ntx = mpg_tile_ntx(h->nx_dst, 64), nty = h->ny_dst;
    size_t lds = sizeof(double) * (65 * (size_t)nlev + 192) + sizeof(int32_t) * 192;
    if (lds > 160 * 1024) {
      mpg_set_error("Regrid(LEV_FAST): %d levels exceed the LDS tile", nlev);
      return MPG_ERR_UNSUPPORTED;
    }
    if (lds > 48 * 1024) MPG_HIP(hipFuncSetAttribute((const void *)k_apply3_lf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_apply3_lf<<<(unsigned)ntx * nty * nfields, 512, lds, s>>>(h->idx.p, h->w.p, src, dst, h->nx_dst, h->ny_dst, h->n_src, nlev, ntx, nty);
  } else if (h->nnz_per_row == 3) {
    // cell-fast (a single level is the same memory in both layouts).  g_a3_staged: -2 = lane-gather kernel only, -1 =
    // per-handle choice by the reuse statistic of the tile lists, >= 0 = that LDS-staged variant (k_apply_lfu.hip)
    int staged = g_a3_staged;
    if (staged == -1) {
      staged = -2;
      if (nlev * nfields >= MPG_STAGE_MIN_LEVELS || h->cf_choice > 0) {   // short bundles: no list build for them
        int rc = mpg_cfu_auto(h, s, &staged);
        if (rc) return rc;
      }
    } else if (staged >= 0) {
      int fits, rc = mpg_cfu_fits(h, staged, s, &fits);
      if (rc) return rc;
      if (!fits) staged = -2;  // hardly any cell shared inside a tile: the lane-gather kernel serves this handle
    }
    if (staged >= 0) {
      int rc = mpg_k_apply3_cfu(h, staged, src, 0, nlev, nfields, dst, 0, false, 1.0, 0.0, s);
      if (rc != MPG_ERR_UNSUPPORTED) return rc;
    }
    int ntx = mpg_tile_ntx(h->nx_dst, A3_TX), nty = (h->ny_dst + 7) / 8;
    k_apply3_cf<<<(unsigned)ntx * nty * nfields, 256, 0, s>>>(h->idx.p, h->w.p, src, dst, h->nx_dst, h->ny_dst, h->n_src, nlev, ntx, nty);
  } else {
    mpg_set_error("Regrid: unsupported handle");
    return MPG_ERR_UNSUPPORTED;
  }
  MPG_HIP(hipGetLastError());
  if (h->n_pole) return mpg_k_pole_fix(h, src, 0, layout, nlev, nfields, dst, 0, 1.0, 0.0, s);
  return MPG_SUCCESS;
}

int mpg_k_rotate(int64_t npts, int nlev, const double *cosa, const double *sina, double *u, double *v, hipStream_t s) {
  if (npts == 0 || nlev == 0) return MPG_SUCCESS;
  k_rotate<<<(unsigned)((npts + 255) / 256), 256, 0, s>>>(npts, nlev, cosa, sina, u, v);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

int mpg_k_pack(const double *src, int64_t n_src, int nlev, const int32_t *ids, int64_t n_ids, double *dst, hipStream_t s) {
  if (n_ids == 0 || nlev == 0) return MPG_SUCCESS;
  k_pack<<<(unsigned)((n_ids + 255) / 256), 256, 0, s>>>(src, n_src, nlev, ids, n_ids, dst);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

// mpg_init loads this translation unit's code object ahead of its first launch (mpg_api.hip: warm_modules)
const void *mpg_anchor_k_apply() { return (const void *)k_apply3_cf; }
