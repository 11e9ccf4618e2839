// K2'' k_apply3_lfu: 3-point Regrid from the level-fastest source ([ncell][nlev], MPAS file order) with the tile's
// UNIQUE source cells staged through LDS.
//
// Why: k_apply3_lf reads three source rows per target point; neighbouring target points share cells (1.4-2.5 points
// per cell on the BASELINE configs), so the same row crosses the L2 -> CU path several times and the kernel ends up
// bound there, not by HBM (4.6 TB/s on C4, 2.9 TB/s on the 6.5 M-point global target).  Here every
// tile of 64 x 4*RPT target points carries the sorted list of the cells its points reference (built once per handle,
// on the device) and each point keeps three 16-bit positions in that list.  Per chunk of LC levels the workgroup
// loads each unique row ONCE (LC consecutive doubles = one 64/128-byte segment per row, lanes along the levels),
// parks it in LDS ([row][LC+1]: odd stride, conflict-free for the column reads) and every thread combines its points
// from LDS; stores are 512-byte non-temporal row segments per level as in the other kernels.
// HBM traffic per tile = unique rows (+ the one-cell halo ring shared with the neighbour tiles) + the destination:
// independent of how the mesh numbers its cells.  Arithmetic = wsum3, bit-identical to the other variants.
#include <limits.h>
#include <string.h>

#include "geom.h"
#include "mpg_internal.h"

#define LFU_TX 64
#define LFU_THREADS 256

__device__ __forceinline__ unsigned lfu_xcd_remap(unsigned lin, unsigned n) {
  unsigned q = n / 8, r = n % 8, xcd = lin % 8, k = lin / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// ---- per-tile unique cell lists -------------------------------------------------------------------------
// One workgroup per tile: the 3*NP cell ids are sorted in LDS (bitonic), duplicates dropped, and each point's three
// ids are replaced by their rank in the tile's list.  FILL = false only counts (-> scan -> FILL = true).
template <int RPT, bool FILL>
__global__ __launch_bounds__(LFU_THREADS) void k_lfu_build(const int32_t *__restrict__ idx, int nx, int ny, int ntx,
                                                           int32_t *__restrict__ ut_count, const int32_t *__restrict__ ut_ptr,
                                                           int32_t *__restrict__ ut_cells, uint16_t *__restrict__ lidx) {
  constexpr int TY = 4 * RPT, NP = LFU_TX * TY, NK = 3 * NP;
  constexpr int SB = NK <= 1024 ? 1024 : (NK <= 2048 ? 2048 : 4096);
  __shared__ int32_t keys[SB];
  __shared__ int32_t part[LFU_THREADS + 1];
  const int64_t P = (int64_t)nx * ny;
  const int tile = blockIdx.x, tx = tile % ntx, ty = tile / ntx, t = threadIdx.x;
  for (int e = t; e < SB; e += LFU_THREADS) {
    int32_t key = INT_MAX;
    if (e < NK) {
      int q = e / NP, pt = e % NP;
      int i = tx * LFU_TX + pt % LFU_TX, j = ty * TY + pt / LFU_TX;
      if (i < nx && j < ny) {
        int32_t c = idx[q * P + (int64_t)j * nx + i];
        if (c >= 0) key = c;
      }
    }
    keys[e] = key;
  }
  __syncthreads();
  for (int k = 2; k <= SB; k <<= 1)
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      for (int e = t; e < SB; e += LFU_THREADS) {
        int partner = e ^ jj;
        if (partner > e) {
          int32_t a = keys[e], b = keys[partner];
          bool up = (e & k) == 0;
          if ((a > b) == up) {
            keys[e] = b;
            keys[partner] = a;
          }
        }
      }
      __syncthreads();
    }
  // unique count: thread t owns the contiguous slice [t*PER, (t+1)*PER)
  constexpr int PER = SB / LFU_THREADS;
  int cnt = 0;
  for (int e = t * PER; e < (t + 1) * PER; ++e) {
    int32_t v = keys[e];
    if (v != INT_MAX && (e == 0 || keys[e - 1] != v)) ++cnt;
  }
  part[t + 1] = cnt;
  if (t == 0) part[0] = 0;
  __syncthreads();
  if (t == 0)
    for (int q = 1; q <= LFU_THREADS; ++q) part[q] += part[q - 1];  // 256 adds, once per tile, set-up only
  __syncthreads();
  const int total = part[LFU_THREADS];
  if (!FILL) {
    if (t == 0) ut_count[tile] = total;
    return;
  }
  // compact in place is unsafe (readers of keys[e-1]): collect this thread's uniques first, then write
  int32_t mine[PER];
  int nm = 0;
  for (int e = t * PER; e < (t + 1) * PER; ++e) {
    int32_t v = keys[e];
    if (v != INT_MAX && (e == 0 || keys[e - 1] != v)) mine[nm++] = v;
  }
  __syncthreads();
  const int base = part[t];
  for (int q = 0; q < nm; ++q) {
    keys[base + q] = mine[q];
    ut_cells[ut_ptr[tile] + base + q] = mine[q];
  }
  __syncthreads();
  for (int pt = t; pt < NP; pt += LFU_THREADS) {
    int i = tx * LFU_TX + pt % LFU_TX, j = ty * TY + pt / LFU_TX;
    if (i >= nx || j >= ny) continue;
    int64_t p = (int64_t)j * nx + i;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      int32_t c = idx[q * P + p];
      int pos = 0xFFFF;
      if (c >= 0) {
        int lo = 0, hi = total - 1;
        while (lo < hi) {
          int mid = (lo + hi) >> 1;
          if (keys[mid] < c) lo = mid + 1;
          else hi = mid;
        }
        pos = lo;
      }
      lidx[q * P + p] = (uint16_t)pos;
    }
  }
}

// ---- Regrid ---------------------------------------------------------------------------------------------
template <int RPT, int LC>
__global__ __launch_bounds__(LFU_THREADS) void k_apply3_lfu(const int32_t *__restrict__ ut_ptr, const int32_t *__restrict__ ut_cells,
                                                            const uint16_t *__restrict__ lidx, const double *__restrict__ w,
                                                            const double *__restrict__ src, double *__restrict__ dst, int nx, int ny,
                                                            int64_t nsrc, int nlev, int ntx, int nty, int nfields, int ut_max) {
  constexpr int TY = 4 * RPT, LS = LC + 1, RPP = LFU_THREADS / LC;  // RPP = rows loaded per pass
  extern __shared__ double lds[];           // rows [ut_max][LS] | cell ids [ut_max]
  int32_t *cells = (int32_t *)(lds + (size_t)ut_max * LS);
  const int64_t P = (int64_t)nx * ny;
  const unsigned ntile = (unsigned)ntx * nty;
  const unsigned lin = lfu_xcd_remap(blockIdx.x, gridDim.x);
  const unsigned tile = lin % ntile;
  const int f = lin / ntile;
  const int tx = tile % ntx, ty = tile / ntx;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int u0 = ut_ptr[tile], nU = ut_ptr[tile + 1] - u0;
  for (int r = t; r < nU; r += LFU_THREADS) cells[r] = ut_cells[u0 + r];

  const int i = tx * LFU_TX + lane;
  const int j0 = ty * TY + wave * RPT;
  int l[RPT][3];
  double ww[RPT][3];
  bool act[RPT], mapped[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    int j = j0 + r;
    act[r] = i < nx && j < ny;
    int64_t p = act[r] ? (int64_t)j * nx + i : 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      l[r][q] = lidx[q * P + p];
      ww[r][q] = w[q * P + p];
    }
    mapped[r] = l[r][0] != 0xFFFF;
#pragma unroll
    for (int q = 0; q < 3; ++q) l[r][q] = mapped[r] ? l[r][q] * LS : 0;
  }
  const double *sf = src + (int64_t)f * nlev * nsrc;
  double *df = dst + (int64_t)f * nlev * P + (int64_t)j0 * nx + i;
  const int lrow = t / LC, llev = t % LC;
  __syncthreads();  // cells[] visible
  for (int k0 = 0; k0 < nlev; k0 += LC) {
    // phase 1: every unique row's LC levels -> LDS, 4 loads in flight per thread
    const bool lev_ok = k0 + llev < nlev;
    for (int rb = lrow; rb < nU; rb += 4 * RPP) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int row = rb + u * RPP;
        v[u] = (row < nU && lev_ok) ? sf[(int64_t)cells[row] * nlev + k0 + llev] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int row = rb + u * RPP;
        if (row < nU) lds[row * LS + llev] = v[u];
      }
    }
    __syncthreads();
    // phase 2: combine from LDS, lanes = consecutive i
    const int kn = min(LC, nlev - k0);
    for (int kk = 0; kk < kn; ++kk) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        double a = lds[l[r][0] + kk], b = lds[l[r][1] + kk], e = lds[l[r][2] + kk];
        double val = wsum3(ww[r][0], a, ww[r][1], b, ww[r][2], e);
        if (act[r]) __builtin_nontemporal_store(mapped[r] ? val : 0.0, df + (int64_t)(k0 + kk) * P + (int64_t)r * nx);
      }
    }
    __syncthreads();
  }
}

// Software-pipelined form: the rows of level chunk c+1 are fetched into registers (NPF per thread) while chunk c is
// combined from LDS and stored, so the global-load latency hides behind the LDS/ALU/store phase instead of sitting
// between two barriers.  Tiles with more than NPF * (256/LC) unique rows load the surplus rows synchronously.
template <int RPT, int LC, int NPF>
__global__ __launch_bounds__(LFU_THREADS) void k_apply3_lfu_p(const int32_t *__restrict__ ut_ptr, const int32_t *__restrict__ ut_cells,
                                                              const uint16_t *__restrict__ lidx, const double *__restrict__ w,
                                                              const double *__restrict__ src, double *__restrict__ dst, int nx, int ny,
                                                              int64_t nsrc, int nlev, int ntx, int nty, int nfields, int ut_max) {
  constexpr int TY = 4 * RPT, LS = LC + 1, RPP = LFU_THREADS / LC;
  extern __shared__ double lds[];
  int32_t *cells = (int32_t *)(lds + (size_t)ut_max * LS);
  const int64_t P = (int64_t)nx * ny;
  const unsigned ntile = (unsigned)ntx * nty;
  const unsigned lin = lfu_xcd_remap(blockIdx.x, gridDim.x);
  const unsigned tile = lin % ntile;
  const int f = lin / ntile;
  const int tx = tile % ntx, ty = tile / ntx;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int u0 = ut_ptr[tile], nU = ut_ptr[tile + 1] - u0;
  for (int r = t; r < nU; r += LFU_THREADS) cells[r] = ut_cells[u0 + r];

  const int i = tx * LFU_TX + lane;
  const int j0 = ty * TY + wave * RPT;
  int l[RPT][3];
  double ww[RPT][3];
  bool act[RPT], mapped[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    int j = j0 + r;
    act[r] = i < nx && j < ny;
    int64_t p = act[r] ? (int64_t)j * nx + i : 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      l[r][q] = lidx[q * P + p];
      ww[r][q] = w[q * P + p];
    }
    mapped[r] = l[r][0] != 0xFFFF;
#pragma unroll
    for (int q = 0; q < 3; ++q) l[r][q] = mapped[r] ? l[r][q] * LS : 0;
  }
  const double *sf = src + (int64_t)f * nlev * nsrc;
  double *df = dst + (int64_t)f * nlev * P + (int64_t)j0 * nx + i;
  const int lrow = t / LC, llev = t % LC;
  __syncthreads();  // cells[] visible
  // this thread's rows: element offsets of (cell, level llev) inside the field, -1 = none
  int64_t roff[NPF];
#pragma unroll
  for (int u = 0; u < NPF; ++u) {
    int row = lrow + u * RPP;
    roff[u] = row < nU ? (int64_t)cells[row] * nlev + llev : -1;
  }
  double pf[NPF];
#pragma unroll
  for (int u = 0; u < NPF; ++u) pf[u] = (roff[u] >= 0 && llev < nlev) ? sf[roff[u]] : 0.0;
  for (int k0 = 0; k0 < nlev; k0 += LC) {
#pragma unroll
    for (int u = 0; u < NPF; ++u)
      if (roff[u] >= 0) lds[(lrow + u * RPP) * LS + llev] = pf[u];
    for (int row = lrow + NPF * RPP; row < nU; row += RPP)  // surplus rows of an unusually large tile
      lds[row * LS + llev] = (k0 + llev < nlev) ? sf[(int64_t)cells[row] * nlev + k0 + llev] : 0.0;
    __syncthreads();
    const int kn1 = k0 + LC;
    if (kn1 < nlev) {
      const bool ok = kn1 + llev < nlev;
#pragma unroll
      for (int u = 0; u < NPF; ++u) pf[u] = (roff[u] >= 0 && ok) ? sf[roff[u] + kn1] : 0.0;
    }
    const int kn = min(LC, nlev - k0);
    for (int kk = 0; kk < kn; ++kk) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        double a = lds[l[r][0] + kk], b = lds[l[r][1] + kk], e = lds[l[r][2] + kk];
        double val = wsum3(ww[r][0], a, ww[r][1], b, ww[r][2], e);
        if (act[r]) __builtin_nontemporal_store(mapped[r] ? val : 0.0, df + (int64_t)(k0 + kk) * P + (int64_t)r * nx);
      }
    }
    __syncthreads();
  }
}

typedef void (*lfu_fn)(const int32_t *, const int32_t *, const uint16_t *, const double *, const double *, double *, int, int, int64_t,
                       int, int, int, int, int);
struct LfuVariant { int rpt, lc; lfu_fn fn; };
static const LfuVariant g_lfu_variants[] = {
    {1, 8, k_apply3_lfu<1, 8>}, {1, 16, k_apply3_lfu<1, 16>}, {2, 8, k_apply3_lfu<2, 8>}, {2, 16, k_apply3_lfu<2, 16>},
    {1, 4, k_apply3_lfu<1, 4>}, {2, 4, k_apply3_lfu<2, 4>},
    // 6..: software-pipelined
    {1, 8, k_apply3_lfu_p<1, 8, 8>}, {1, 16, k_apply3_lfu_p<1, 16, 12>}, {2, 8, k_apply3_lfu_p<2, 8, 16>},
    {2, 16, k_apply3_lfu_p<2, 16, 16>}, {1, 8, k_apply3_lfu_p<1, 8, 12>}, {1, 16, k_apply3_lfu_p<1, 16, 16>},
};
int mpg_lfu_num_variants() { return (int)(sizeof(g_lfu_variants) / sizeof(g_lfu_variants[0])); }

static int lfu_build(mpg_handle_s *h, int rpt, hipStream_t s) {
  if (h->ut_rpt == rpt) return MPG_SUCCESS;
  int rc;
  h->ut_ptr.free();
  h->ut_cells.free();
  h->ut_rpt = 0;
  const int ty = 4 * rpt;
  const int ntx = (h->nx_dst + LFU_TX - 1) / LFU_TX, nty = (h->ny_dst + ty - 1) / ty;
  const int64_t ntile = (int64_t)ntx * nty;
  TmpBuf<int32_t> count;
  if ((rc = count.alloc(ntile + 1)) || (rc = h->ut_ptr.alloc(ntile + 1))) return rc;
  if (!h->lidx.p && (rc = h->lidx.alloc(3 * (size_t)h->n_dst))) return rc;
  MPG_HIP(hipMemsetAsync(count.p, 0, sizeof(int32_t) * (ntile + 1), s));
  if (rpt == 1) k_lfu_build<1, false><<<(unsigned)ntile, LFU_THREADS, 0, s>>>(h->idx.p, h->nx_dst, h->ny_dst, ntx, count.p, nullptr, nullptr, nullptr);
  else k_lfu_build<2, false><<<(unsigned)ntile, LFU_THREADS, 0, s>>>(h->idx.p, h->nx_dst, h->ny_dst, ntx, count.p, nullptr, nullptr, nullptr);
  MPG_HIP(hipGetLastError());
  std::vector<int32_t> hc((size_t)ntile + 1), hp((size_t)ntile + 1);
  MPG_HIP(hipMemcpyAsync(hc.data(), count.p, sizeof(int32_t) * (ntile + 1), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  int64_t tot = 0;
  int mx = 0;
  for (int64_t q = 0; q < ntile; ++q) {
    hp[q] = (int32_t)tot;
    tot += hc[q];
    mx = hc[q] > mx ? hc[q] : mx;
  }
  hp[ntile] = (int32_t)tot;
  if (tot >= 0x7fffffff) {
    mpg_set_error("tile cell lists exceed 2^31 entries");
    return MPG_ERR_OVERFLOW;
  }
  MPG_HIP(hipMemcpyAsync(h->ut_ptr.p, hp.data(), sizeof(int32_t) * (ntile + 1), hipMemcpyHostToDevice, s));
  if ((rc = h->ut_cells.alloc((size_t)tot + 1))) return rc;
  if (rpt == 1) k_lfu_build<1, true><<<(unsigned)ntile, LFU_THREADS, 0, s>>>(h->idx.p, h->nx_dst, h->ny_dst, ntx, nullptr, h->ut_ptr.p, h->ut_cells.p, h->lidx.p);
  else k_lfu_build<2, true><<<(unsigned)ntile, LFU_THREADS, 0, s>>>(h->idx.p, h->nx_dst, h->ny_dst, ntx, nullptr, h->ut_ptr.p, h->ut_cells.p, h->lidx.p);
  MPG_HIP(hipGetLastError());
  MPG_HIP(hipStreamSynchronize(s));
  h->ut_rpt = rpt;
  h->ut_max = mx;
  h->ut_total = tot;
  return MPG_SUCCESS;
}

// Which level-fast kernel serves this handle?  Measured on MI355X (profiles/r01_sweep_lfu.txt), 4 fields x 55 levels:
//   target points per source cell   row-gather k_apply3_lf   LDS-staged (variant 11)
//   1.4  (C4, 3 M cells)            4.76 TB/s                3.3 TB/s
//   2.9  (C2, 655 k cells)          2.61 TB/s                4.53 TB/s
//   2.5  (C5, global lat-lon)       2.80 TB/s                4.72 TB/s
// Staging pays when a staged row is referenced often enough; the statistic that separates the cases is
// reuse = 3 * n_dst / sum(unique cells per tile): 2.5 on C4, 5-6 on C2 / C5.
#define LFU_AUTO_VARIANT 11
#define LFU_AUTO_MIN_REUSE 3.5f
int mpg_lfu_auto(mpg_handle_s *h, hipStream_t s, int *lfu_variant) {
  if (h->lf_choice == 0) {
    int rc = lfu_build(h, g_lfu_variants[LFU_AUTO_VARIANT].rpt, s);
    if (rc) return rc;
    h->lf_reuse = h->ut_total > 0 ? 3.0f * (float)h->n_dst / (float)h->ut_total : 0.f;
    h->lf_choice = h->lf_reuse >= LFU_AUTO_MIN_REUSE ? 1 : -1;
    if (h->lf_choice < 0) {  // not needed: give the memory back
      h->ut_ptr.free();
      h->ut_cells.free();
      h->lidx.free();
      h->ut_rpt = 0;
    }
  }
  *lfu_variant = h->lf_choice > 0 ? LFU_AUTO_VARIANT : -1;
  return MPG_SUCCESS;
}

int mpg_k_apply3_lfu(mpg_handle_s *h, int variant, const double *src, int nlev, int nfields, double *dst, hipStream_t s) {
  const LfuVariant &v = g_lfu_variants[variant];
  int rc = lfu_build(h, v.rpt, s);
  if (rc) return rc;
  const int ty = 4 * v.rpt;
  const int ntx = (h->nx_dst + LFU_TX - 1) / LFU_TX, nty = (h->ny_dst + ty - 1) / ty;
  const size_t um = h->ut_max > 0 ? h->ut_max : 1;  // unmapped points read row 0
  size_t lds = sizeof(double) * um * (v.lc + 1) + sizeof(int32_t) * um + 16;
  if (lds > 160 * 1024) {
    mpg_set_error("Regrid(LEV_FAST, staged): %d unique cells per tile exceed the LDS", h->ut_max);
    return MPG_ERR_UNSUPPORTED;
  }
  if (lds > 48 * 1024) MPG_HIP(hipFuncSetAttribute((const void *)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  v.fn<<<(unsigned)ntx * nty * nfields, LFU_THREADS, lds, s>>>(h->ut_ptr.p, h->ut_cells.p, h->lidx.p, h->w.p, src, dst, h->nx_dst, h->ny_dst,
                                                              h->n_src, nlev, ntx, nty, nfields, (int)um);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}
