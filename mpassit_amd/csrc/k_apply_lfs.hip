// K2''' k_apply3_lfs: 3-point Regrid from the level-fastest source ([ncell][nlev], the order of the MPAS history file:
// input_data.F90:630 reads dummy3(nz,nCells,1)) with the COMPLETE rows of a tile's unique source cells resident in LDS.
//   * a tile is 64 x TY target points with the sorted list of the cells it references (the lists of k_apply_lfu.hip:
//     built once per handle, three 16-bit ranks per point); every source row crosses L2 -> CU once per tile (the
//     row-gather kernels fetch ~2.3 rows per referenced cell and tile on C4), whatever the cell numbering;
//   * phase 1: all row loads of the tile back to back, NB in flight per lane -- float32 rows as one unaligned 8-byte load
//     per lane and half-wave (a 55-level row = 220 bytes = 28 lanes), float64 rows one element per lane -- then parked in
//     LDS in the SOURCE element type, row stride odd in words (conflict-free column reads).  No divergent branch: rows
//     past the end of the list and lanes past the end of a row are clamped onto valid ones and re-write identical values;
//   * phase 2: thread = (target point, level group).  The level group is wave-uniform, so the level loop is scalar, the
//     destination plane is a scalar base with one 32-bit lane offset, and the LDS addresses are a lane base + immediate
//     offsets.  Per output: 1.5 LDS-read instructions (ds_read2), three widenings, wsum3 (the FMA pattern of every other
//     variant: bit-identical results), the writer's affine epilogue (T - 300, PHB * 9.81, NF90_FLOAT narrowing) and one
//     non-temporal store; a wave stores 64 consecutive points of one level.  Unmapped points carry zero weights and
//     point at an all-zero LDS row instead of being selected away.
// Measured (round 2, profiles/r02_lfs_*.txt, float32 in / out, 13 fields): C5 6.82 ms and the 655 k-cell configurations
// 1.62 / 2.20 ms -- 2-6 % faster than the level-chunked staged kernel; C4 3.4-3.6 ms against 3.1-3.3 ms of the
// row-gather kernel k_apply3_lf_f32x2 (later replaced as the default by k_apply3_lf_f32m, k_apply_typed.hip: 2.6 ms on the
// same workload with linear aligned tiles, 32-bit row offsets and 8 workgroups per CU).  MODE 1-7 are the ablations that show
// why: every phase alone is fast (stores alone 6.3 TB/s, row gather alone 7 TB/s), but a workgroup
// lives 7.5 us -- 2.7 us of dependent prologue loads, 2.9 us until its rows have landed, 1.1 us combine, 0.5 us store
// acknowledgement -- and LDS lets only ~5 of them share a CU; with 4-byte elements that latency chain, not bytes or
// instruction issue (halving the instruction count changed nothing), is what sets the time.
#include <type_traits>

#include "geom.h"
#include "mpg_internal.h"

#define LFS_THREADS 256

typedef float lfs_f32x2 __attribute__((ext_vector_type(2), aligned(4)));
typedef float lfs_f32x4 __attribute__((ext_vector_type(4), aligned(4)));

// MODE 0: the kernel.  1-4: timing ablations (plain stores / no stores / no loads / neither), never on a product path.
template <typename TS, typename TD, int TY, int NH, int NB, int MODE>
__global__ __launch_bounds__(LFS_THREADS) void k_apply3_lfs(const int32_t *__restrict__ ut_ptr, const int32_t *__restrict__ ut_cells,
                                                            const uint16_t *__restrict__ lidx, const double *__restrict__ w,
                                                            const TS *__restrict__ src, TD *__restrict__ dst, int nx, int ny, int talign, int64_t nsrc,
                                                            int nlev, int rs, int ntx, int nty, int ut_max, double scale, double offset) {
  constexpr int NP = 64 * TY;                       // target points per tile
  constexpr int UN = 4;                             // levels per trip of the combine loop
  static_assert(NP * NH == LFS_THREADS, "one (point, level group) item per thread");
  extern __shared__ double lds_raw[];
  TS *rows = (TS *)lds_raw;                         // [ut_max + 1][rs]; row ut_max is all zero (unmapped points)
  int32_t *cells = (int32_t *)(rows + (size_t)(ut_max + 1) * rs);
  const int64_t P = (int64_t)nx * ny;
  const unsigned ntile = (unsigned)ntx * nty;
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned tile = lin % ntile;
  const int f = lin / ntile;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  unsigned long long stamp[6];
  if constexpr (MODE == 5) stamp[0] = __builtin_amdgcn_s_memtime();
  const int u0 = ut_ptr[tile], nU = ut_ptr[tile + 1] - u0;
  for (int r = t; r < nU; r += LFS_THREADS) cells[r] = ut_cells[u0 + r];
  for (int e = t; e < rs; e += LFS_THREADS) rows[(size_t)ut_max * rs + e] = (TS)0;
  // this thread's target point: ranks and weights stay in registers for all levels
  const int tx = tile % ntx, ty = tile / ntx;
  const int pt = t % NP;
  const int lg = __builtin_amdgcn_readfirstlane(t / NP);      // level group: uniform over the wave
  const int j = ty * TY + (pt >> 6), i = tx * 64 + (pt & 63) - mpg_tile_shift(j, nx, talign);
  const bool act = i >= 0 && i < nx && j < ny;
  const unsigned off = act ? (unsigned)(j * nx + i) : 0u;
  int l0 = lidx[off], l1 = lidx[P + off], l2 = lidx[2 * P + off];
  double w0 = w[off], w1 = w[P + off], w2 = w[2 * P + off];
  {
    const bool mapped = l0 != 0xFFFF;
    l0 = mapped ? l0 * rs : ut_max * rs;
    l1 = mapped ? l1 * rs : ut_max * rs;
    l2 = mapped ? l2 * rs : ut_max * rs;
    w0 = mapped ? w0 : 0.0;
    w1 = mapped ? w1 : 0.0;
    w2 = mapped ? w2 : 0.0;
  }
  const TS *sf = src + (int64_t)f * nlev * nsrc;
  __syncthreads();  // cells[] visible
  if constexpr (MODE == 5) {   // stamp 1: cell list staged, ranks and weights landed
    __builtin_amdgcn_s_waitcnt(0);
    stamp[1] = __builtin_amdgcn_s_memtime();
  }
  // ---- phase 1: all rows of the tile, NB loads in flight per lane, branch-free -----------------------------------------
  if (MODE != 3 && MODE != 4 && MODE != 6 && nU > 0) {
    if constexpr (sizeof(TS) == 4) {
      // half-wave = one row; lane sl holds levels l + 2 sl, l + 2 sl + 1 (one unaligned 8-byte load).  Lanes at or past the
      // end of the row are clamped onto its last two levels, rows past the end of the list onto the last row: they fetch and
      // re-write values another lane writes identically.
      const int rsub = 2 * wave + (lane >> 5), sl = lane & 31;
      for (int lv = 0; lv < nlev; lv += 64) {
        const int base = min(lv + 2 * sl, nlev - 2);
        for (int b0 = 0; b0 < nU; b0 += 8 * NB) {
          lfs_f32x2 v[NB];
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            const int row = min(b0 + 8 * u + rsub, nU - 1);
            v[u] = *(const lfs_f32x2 *)(sf + (int64_t)cells[row] * nlev + base);
          }
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            const int row = min(b0 + 8 * u + rsub, nU - 1);
            float *d = (float *)rows + row * rs + base;
            d[0] = v[u].x;
            d[1] = v[u].y;
          }
        }
      }
    } else {
      for (int lv = 0; lv < nlev; lv += 64) {
        const int k = min(lv + lane, nlev - 1);
        for (int b0 = 0; b0 < nU; b0 += 4 * NB) {
          TS v[NB];
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            const int row = min(b0 + 4 * u + wave, nU - 1);
            v[u] = sf[(int64_t)cells[row] * nlev + k];
          }
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            const int row = min(b0 + 4 * u + wave, nU - 1);
            rows[row * rs + k] = v[u];
          }
        }
      }
    }
  }
  if constexpr (MODE == 5) stamp[2] = __builtin_amdgcn_s_memtime();   // stamp 2: this wave's rows landed and parked
  __syncthreads();
  if constexpr (MODE == 5) stamp[3] = __builtin_amdgcn_s_memtime();   // stamp 3: all rows of the tile in LDS
  // ---- phase 2: combine from LDS, epilogue, store ------------------------------------------------------------------------
  if (MODE != 5 && !act) return;
  if constexpr (MODE == 7) return;                  // microbenchmark: row gather into LDS only
  if constexpr (MODE == 6) {                        // microbenchmark: the store pattern only (no loads, no LDS reads, no arithmetic)
    TD *pl = dst + (int64_t)f * nlev * P;
    for (int k = lg; k < nlev; k += NH) __builtin_nontemporal_store((TD)k, pl + (int64_t)k * P + off);
    return;
  }
  const TS *r0 = rows + l0, *r1 = rows + l1, *r2 = rows + l2;
  TD *plane = dst + (int64_t)f * nlev * P;          // scalar; the lane adds its 32-bit point offset
  auto put = [&](TD v, int kk) {
    TD *d = plane + (int64_t)kk * P;
    if constexpr (MODE == 1) d[off] = v;                                            // ablation: plain stores
    else if constexpr (MODE == 2 || MODE == 4) { if (v == (TD)1.2345e30) d[off] = v; }  // ablation: no stores
    else if constexpr (MODE == 5) { if (act) __builtin_nontemporal_store(v, d + off); }
    else __builtin_nontemporal_store(v, d + off);
  };
  int k = lg;
  for (; k + (UN - 1) * NH < nlev; k += NH * UN) {
    TS a[UN], b[UN], c[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      a[u] = r0[k + u * NH];
      b[u] = r1[k + u * NH];
      c[u] = r2[k + u * NH];
    }
    TD o[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) o[u] = (TD)fma(wsum3(w0, (double)a[u], w1, (double)b[u], w2, (double)c[u]), scale, offset);
#pragma unroll
    for (int u = 0; u < UN; ++u) put(o[u], k + u * NH);
  }
  for (; k < nlev; k += NH) put((TD)fma(wsum3(w0, (double)r0[k], w1, (double)r1[k], w2, (double)r2[k]), scale, offset), k);
  if constexpr (MODE == 5) {   // stamps 4 / 5: stores issued / stores acknowledged; written BEHIND the destination (tools/lfs_stamps.py)
    stamp[4] = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0);
    stamp[5] = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
      unsigned long long *o = (unsigned long long *)(dst + (int64_t)gridDim.x / ntile * nlev * P) + ((size_t)blockIdx.x * 4 + wave) * 8;
      for (int q = 0; q < 6; ++q) o[q] = stamp[q];
      o[6] = (unsigned long long)nU;
      o[7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID: where this wave ran
    }
  }
}

struct LfsShape { int ty, nh; };
// lf_variant 300 + index
static const LfsShape g_lfs_shapes[] = {{1, 4}, {2, 2}, {4, 1}};
#define LFS_NSHAPES 3
int mpg_lfs_num_variants() { return 80; }  // 0-2 shapes; 10 * mode + shape: timing ablations of shapes 0 / 1 (float32 only)

template <typename TS, typename TD, int TY, int NH, int MODE = 0>
static int launch_lfs(mpg_handle_s *h, const void *src, int nlev, int nfields, void *dst, double scale, double offset, size_t lds_cap,
                      hipStream_t s) {
  constexpr int NB = 16;
  const int ntx = mpg_tile_ntx(h->nx_dst, 64, h->ut_align), nty = (h->ny_dst + TY - 1) / TY;
  const size_t um = h->ut_max > 0 ? h->ut_max : 1;
  const int rs = nlev | 1;  // row stride in elements, odd: a column read walks all banks (float64: all bank pairs)
  const size_t lds = sizeof(TS) * (um + 1) * rs + sizeof(int32_t) * um + 16;
  if (lds > lds_cap || (um + 1) * (size_t)rs >= (1u << 30) || (int64_t)h->nx_dst * h->ny_dst >= (int64_t)1 << 31) return MPG_ERR_UNSUPPORTED;
  auto fn = k_apply3_lfs<TS, TD, TY, NH, NB, MODE>;
  if (lds > 48 * 1024) MPG_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  fn<<<(unsigned)ntx * nty * nfields, LFS_THREADS, lds, s>>>(h->ut_ptr.p, h->ut_cells.p, h->lidx.p, h->w.p, (const TS *)src, (TD *)dst, h->nx_dst,
                                                           h->ny_dst, h->ut_align, h->n_src, nlev, rs, ntx, nty, (int)um, scale, offset);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

template <typename TS, typename TD>
static int launch_lfs_shape(mpg_handle_s *h, int shape, const void *src, int nlev, int nfields, void *dst, double scale, double offset,
                            size_t lds_cap, hipStream_t s) {
  if (shape >= 10) {   // ablations (float32 in / out only): 10 * mode + shape
    if (sizeof(TS) != 4 || sizeof(TD) != 4) return MPG_ERR_UNSUPPORTED;
    switch (shape) {
      case 10: return launch_lfs<float, float, 1, 4, 1>(h, src, nlev, nfields, dst, scale, offset, lds_cap, s);
      case 11: return launch_lfs<float, float, 2, 2, 1>(h, src, nlev, nfields, dst, scale, offset, lds_cap, s);
      case 20: return launch_lfs<float, float, 1, 4, 2>(h, src, nlev, nfields, dst, scale, offset, lds_cap, s);
      case 21: return launch_lfs<float, float, 2, 2, 2>(h, src, nlev, nfields, dst, scale, offset, lds_cap, s);
      case 30: return launch_lfs<float, float, 1, 4, 3>(h, src, nlev, nfields, dst, scale, offset, lds_cap, s);
      case 31: return launch_lfs<float, float, 2, 2, 3>(h, src, nlev, nfields, dst, scale, offset, lds_cap, s);
      case 40: return launch_lfs<float, float, 1, 4, 4>(h, src, nlev, nfields, dst, scale, offset, lds_cap, s);
      case 41: return launch_lfs<float, float, 2, 2, 4>(h, src, nlev, nfields, dst, scale, offset, lds_cap, s);
      case 60: return launch_lfs<float, float, 1, 4, 6>(h, src, nlev, nfields, dst, scale, offset, lds_cap, s);   // stores only
      case 61: return launch_lfs<float, float, 2, 2, 6>(h, src, nlev, nfields, dst, scale, offset, lds_cap, s);
      case 70: return launch_lfs<float, float, 1, 4, 7>(h, src, nlev, nfields, dst, scale, offset, lds_cap, s);   // row gather only
      case 71: return launch_lfs<float, float, 2, 2, 7>(h, src, nlev, nfields, dst, scale, offset, lds_cap, s);
      case 50: return launch_lfs<float, float, 1, 4, 5>(h, src, nlev, nfields, dst, scale, offset, lds_cap, s);   // timestamps
      default: return MPG_ERR_UNSUPPORTED;
    }
  }
  switch (shape) {
    case 0: return launch_lfs<TS, TD, 1, 4>(h, src, nlev, nfields, dst, scale, offset, lds_cap, s);
    case 1: return launch_lfs<TS, TD, 2, 2>(h, src, nlev, nfields, dst, scale, offset, lds_cap, s);
    default: return launch_lfs<TS, TD, 4, 1>(h, src, nlev, nfields, dst, scale, offset, lds_cap, s);
  }
}

// Rows-resident level-fast Regrid with tiles of 64 x g_lfs_shapes[shape].ty points.  -> MPG_ERR_UNSUPPORTED when the
// tile's rows do not fit `lds_cap` bytes of LDS (the caller falls back to another kernel) or nlev < 2.
int mpg_k_apply3_lfs(mpg_handle_s *h, int shape, const void *src, int src_f32, int nlev, int nfields, void *dst, int dst_f32, double scale,
                     double offset, size_t lds_cap, hipStream_t s) {
  if (shape < 0 || shape >= mpg_lfs_num_variants() || nlev < 2 || (shape % 10) >= LFS_NSHAPES) return MPG_ERR_UNSUPPORTED;
  int rc = mpg_lfu_build_shape(h, 64, g_lfs_shapes[shape % 10].ty, s);
  if (rc) return rc;
  if (src_f32 && dst_f32) return launch_lfs_shape<float, float>(h, shape, src, nlev, nfields, dst, scale, offset, lds_cap, s);
  if (src_f32) return launch_lfs_shape<float, double>(h, shape, src, nlev, nfields, dst, scale, offset, lds_cap, s);
  if (dst_f32) return launch_lfs_shape<double, float>(h, shape, src, nlev, nfields, dst, scale, offset, lds_cap, s);
  return launch_lfs_shape<double, double>(h, shape, src, nlev, nfields, dst, scale, offset, lds_cap, s);
}
