// K2'' k_apply3_lfu: 3-point Regrid from the level-fastest source ([ncell][nlev], MPAS file order) with the tile's
// UNIQUE source cells staged through LDS.
//
// Why: k_apply3_lf reads three source rows per target point; neighbouring target points share cells (1.4-2.9 points
// per cell on the BASELINE configs), so the same row crosses the L2 -> CU path several times and that kernel ends up
// latency/issue bound, not HBM bound (2.6-2.8 TB/s on the 655 k-cell and global configs).  Here every tile of
// TXU x TY target points (256*RPT points) carries the sorted list of the cells its points reference (built once per
// handle, on the device) and each point keeps three 16-bit positions in that list.  Per chunk of LC levels the
// workgroup loads each unique row ONCE (LC consecutive doubles = one 64/128-byte segment per row, lanes along the
// levels), parks it in LDS ([row][LC+1]: odd stride, conflict-free column reads) and every thread combines its
// points from LDS; stores are >= 128-byte non-temporal row segments per level as in the other kernels.  The
// pipelined form fetches chunk c+1 into registers while chunk c is combined and stored.
// HBM traffic per tile = unique rows (+ the one-cell ring shared with the neighbour tiles) + the destination,
// independent of how the mesh numbers its cells; compact tiles (32 x 8, 16 x 16) keep the ring small.
// Arithmetic = wsum3, bit-identical to the other variants.
#include <limits.h>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "geom.h"
#include <utility>

#include "mpg_internal.h"

#define LFU_THREADS 256
#define LFU_SORT 4096   // sort buffer: 3 ids x (at most) 1024 points, padded to a power of two


// ---- per-tile unique cell lists (set-up, runtime tile shape) --------------------------------------------------
// One workgroup per tile of txu x tyu points: the 3*np cell ids are sorted in LDS (bitonic), duplicates dropped,
// and each point's three ids are replaced by their rank in the tile's list.  FILL = false only counts.
template <bool FILL, int SB = LFU_SORT>
__global__ __launch_bounds__(LFU_THREADS) void k_lfu_build(const int32_t *__restrict__ idx, int nx, int ny, int talign, int txu, int tyu, int ntx,
                                                           int32_t *__restrict__ ut_count, const int32_t *__restrict__ ut_ptr,
                                                           int32_t *__restrict__ ut_cells, uint16_t *__restrict__ lidx,
                                                           unsigned long long *__restrict__ line_count) {
  constexpr int PER = SB / LFU_THREADS;
  __shared__ int32_t keys[SB];
  __shared__ int32_t part[LFU_THREADS + 1];
  const int np = txu * tyu, nk = 3 * np;
  const int64_t P = (int64_t)nx * ny;
  const int tile = blockIdx.x, tx = tile % ntx, ty = tile / ntx, t = threadIdx.x;
  for (int e = t; e < SB; e += LFU_THREADS) {
    int32_t key = INT_MAX;
    if (e < nk) {
      int q = e / np, pt = e % np;
      int j = ty * tyu + pt / txu, i = tx * txu + pt % txu - mpg_tile_shift(j, nx, talign);
      if (i >= 0 && i < nx && j < ny) {
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
  // thread t owns the contiguous slice [t*PER, (t+1)*PER): collect its first occurrences
  int32_t mine[PER];
  int nm = 0;
  for (int e = t * PER; e < (t + 1) * PER; ++e) {
    int32_t v = keys[e];
    if (v != INT_MAX && (e == 0 || keys[e - 1] != v)) mine[nm++] = v;
  }
  part[t + 1] = nm;
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
  const int base = part[t];   // every read of keys[] happened before the barriers above: compact in place
  for (int q = 0; q < nm; ++q) {
    keys[base + q] = mine[q];
    ut_cells[ut_ptr[tile] + base + q] = mine[q];
  }
  __syncthreads();
  // locality statistic: distinct groups of 16 consecutive ids (= 128-byte lines of a cell-fast float64 field) in the list
  if (line_count) {
    int nl = 0;
    for (int e = t; e < total; e += LFU_THREADS) nl += e == 0 || (keys[e] >> 4) != (keys[e - 1] >> 4);
    for (int o = 32; o > 0; o >>= 1) nl += __shfl_down(nl, o);
    if ((t & 63) == 0 && nl) atomicAdd(line_count, (unsigned long long)nl);
  }
  for (int pt = t; pt < np; pt += LFU_THREADS) {
    int j = ty * tyu + pt / txu, i = tx * txu + pt % txu - mpg_tile_shift(j, nx, talign);
    if (i < 0 || i >= nx || j >= ny) continue;
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
// Thread t serves points pt = t + 256*r (r < RPT) of the tile in row-major order: a wave covers 64 consecutive
// points = 64/TXU rows of TXU points (512-, 256- or 128-byte store segments).
template <int TXU, int RPT, int NT = LFU_THREADS>
struct LfuPoints {
  int l[RPT][3];
  double ww[RPT][3];
  bool act[RPT], mapped[RPT];
  int64_t off[RPT];  // j*nx + i
  __device__ __forceinline__ void load(const uint16_t *__restrict__ lidx, const double *__restrict__ w, int nx, int ny, int talign, int tx, int ty, int LS) {
    constexpr int TY = NT * RPT / TXU;
    const int64_t P = (int64_t)nx * ny;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      int pt = (int)threadIdx.x + NT * r;
      int j = ty * TY + pt / TXU, i = tx * TXU + pt % TXU - mpg_tile_shift(j, nx, talign);
      act[r] = i >= 0 && i < nx && j < ny;
      off[r] = act[r] ? (int64_t)j * nx + i : 0;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        l[r][q] = lidx[q * P + off[r]];
        ww[r][q] = w[q * P + off[r]];
      }
      mapped[r] = l[r][0] != 0xFFFF;
#pragma unroll
      for (int q = 0; q < 3; ++q) l[r][q] = mapped[r] ? l[r][q] * LS : 0;
    }
  }
};

// two-phase form (load chunk, barrier, combine, barrier): the reference implementation of the scheme
template <int TXU, int RPT, int LC>
__global__ __launch_bounds__(LFU_THREADS) void k_apply3_lfu(const int32_t *__restrict__ ut_ptr, const int32_t *__restrict__ ut_cells,
                                                            const uint16_t *__restrict__ lidx, const double *__restrict__ w,
                                                            const double *__restrict__ src, double *__restrict__ dst, int nx, int ny, int talign,
                                                            int64_t nsrc, int nlev, int ntx, int nty, int nfields, int ut_max) {
  constexpr int LS = LC + 1, RPP = LFU_THREADS / LC;  // RPP = rows loaded per pass
  extern __shared__ double lds[];                     // rows [ut_max][LS] | cell ids [ut_max]
  int32_t *cells = (int32_t *)(lds + (size_t)ut_max * LS);
  const int64_t P = (int64_t)nx * ny;
  const unsigned ntile = (unsigned)ntx * nty;
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned tile = lin % ntile;
  const int f = lin / ntile;
  const int t = threadIdx.x;
  const int u0 = ut_ptr[tile], nU = ut_ptr[tile + 1] - u0;
  for (int r = t; r < nU; r += LFU_THREADS) cells[r] = ut_cells[u0 + r];
  LfuPoints<TXU, RPT> pts;
  pts.load(lidx, w, nx, ny, talign, tile % ntx, tile / ntx, LS);
  const double *sf = src + (int64_t)f * nlev * nsrc;
  double *df = dst + (int64_t)f * nlev * P;
  const int lrow = t / LC, llev = t % LC;
  __syncthreads();  // cells[] visible
  for (int k0 = 0; k0 < nlev; k0 += LC) {
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
    const int kn = min(LC, nlev - k0);
    for (int kk = 0; kk < kn; ++kk) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        double a = lds[pts.l[r][0] + kk], b = lds[pts.l[r][1] + kk], e = lds[pts.l[r][2] + kk];
        double val = wsum3(pts.ww[r][0], a, pts.ww[r][1], b, pts.ww[r][2], e);
        if (pts.act[r]) __builtin_nontemporal_store(pts.mapped[r] ? val : 0.0, df + (int64_t)(k0 + kk) * P + pts.off[r]);
      }
    }
    __syncthreads();
  }
}

// Software-pipelined form: the rows of level chunk c+1 are fetched into registers (NPF per thread) while chunk c is
// combined from LDS and stored, so the global-load latency hides behind the LDS/ALU/store phase instead of sitting
// between two barriers.  Tiles with more than NPF * (256/LC) unique rows load the surplus rows synchronously.
template <int TXU, int RPT, int LC, int NPF>
__global__ __launch_bounds__(LFU_THREADS) void k_apply3_lfu_p(const int32_t *__restrict__ ut_ptr, const int32_t *__restrict__ ut_cells,
                                                              const uint16_t *__restrict__ lidx, const double *__restrict__ w,
                                                              const double *__restrict__ src, double *__restrict__ dst, int nx, int ny, int talign,
                                                              int64_t nsrc, int nlev, int ntx, int nty, int nfields, int ut_max) {
  constexpr int LS = LC + 1, RPP = LFU_THREADS / LC;
  extern __shared__ double lds[];
  int32_t *cells = (int32_t *)(lds + (size_t)ut_max * LS);
  const int64_t P = (int64_t)nx * ny;
  const unsigned ntile = (unsigned)ntx * nty;
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned tile = lin % ntile;
  const int f = lin / ntile;
  const int t = threadIdx.x;
  const int u0 = ut_ptr[tile], nU = ut_ptr[tile + 1] - u0;
  for (int r = t; r < nU; r += LFU_THREADS) cells[r] = ut_cells[u0 + r];
  LfuPoints<TXU, RPT> pts;
  pts.load(lidx, w, nx, ny, talign, tile % ntx, tile / ntx, LS);
  const double *sf = src + (int64_t)f * nlev * nsrc;
  double *df = dst + (int64_t)f * nlev * P;
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
        double a = lds[pts.l[r][0] + kk], b = lds[pts.l[r][1] + kk], e = lds[pts.l[r][2] + kk];
        double val = wsum3(pts.ww[r][0], a, pts.ww[r][1], b, pts.ww[r][2], e);
        if (pts.act[r]) __builtin_nontemporal_store(pts.mapped[r] ? val : 0.0, df + (int64_t)(k0 + kk) * P + pts.off[r]);
      }
    }
    __syncthreads();
  }
}

// ---- the same staging for the CELL-fastest source ([nlev][ncell], the reference's in-memory order) -------------------
// k_apply3_cf issues three 64-lane gathers per target row and level; with 2.9 target points per cell (C2) two thirds of
// those lanes fetch a value a neighbour lane fetches too, and the kernel needs as long as on C4 although it moves 40 %
// fewer bytes (5.1 vs 4.9 ms: bound by gather lanes, not by HBM).  Staged: per chunk of LC levels the workgroup loads
// the tile's unique cells once (lanes along the sorted cell list: neighbouring ids, coalesced) into LDS [LC][NUP] and
// the points combine from there.  Same tile lists, same wsum3 arithmetic; chunk c+1 is prefetched into registers
// (NPF = LC * ceil(NUP/256) values) while chunk c is combined and stored.
template <int TXU, int RPT, int LC, int NPF, int NT = LFU_THREADS>
__global__ __launch_bounds__(NT) void k_apply3_cfu_p(const int32_t *__restrict__ ut_ptr, const int32_t *__restrict__ ut_cells,
                                                              const uint16_t *__restrict__ lidx, const double *__restrict__ w,
                                                              const double *__restrict__ src, double *__restrict__ dst, int nx, int ny, int talign,
                                                              int64_t nsrc, int nlev, int ntx, int nty, int nfields, int ut_max) {
  constexpr int UPT = NPF / LC;                 // unique cells per thread held in registers
  extern __shared__ double lds[];               // [LC][nup]
  const int nup = ut_max;
  const int64_t P = (int64_t)nx * ny;
  const unsigned ntile = (unsigned)ntx * nty;
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned tile = band_order(lin % ntile, ntx, nty, (unsigned)nfields >> 24);
  // A workgroup serves `fpw` consecutive fields of its tile (packed into the upper half of `nfields` by the launcher): the
  // bundle is stored [field][level][cell] / [field][level][point], so fields f0 .. f0+fpw-1 are simply fpw * nlev consecutive
  // "levels" -- the chunk pipeline runs across the field boundaries, and the tile's list, ranks and weights are fetched
  // once per fpw fields instead of once per field.
  const int fpw = max(1, (nfields >> 16) & 0xff), nf = nfields & 0xffff;   // bits 24..31: tile band (band_order)
  const int f = (int)(lin / ntile) * fpw;
  const int nlev_all = nlev;
  nlev = min(fpw, nf - f) * nlev_all;
  const int t = threadIdx.x;
  const int u0 = ut_ptr[tile], nU = ut_ptr[tile + 1] - u0;
  LfuPoints<TXU, RPT, NT> pts;
  pts.load(lidx, w, nx, ny, talign, tile % ntx, tile / ntx, 1);
  const double *sf = src + (int64_t)f * nlev_all * nsrc;
  double *df = dst + (int64_t)f * nlev_all * P;
  int32_t cell[UPT];
#pragma unroll
  for (int u = 0; u < UPT; ++u) {
    int q = t + NT * u;
    cell[u] = q < nU ? ut_cells[u0 + q] : -1;
  }
  double pf[NPF];
#pragma unroll
  for (int lv = 0; lv < LC; ++lv)
#pragma unroll
    for (int u = 0; u < UPT; ++u) pf[lv * UPT + u] = (cell[u] >= 0 && lv < nlev) ? sf[(int64_t)lv * nsrc + cell[u]] : 0.0;
  for (int k0 = 0; k0 < nlev; k0 += LC) {
#pragma unroll
    for (int lv = 0; lv < LC; ++lv)
#pragma unroll
      for (int u = 0; u < UPT; ++u)
        if (cell[u] >= 0) lds[lv * nup + t + NT * u] = pf[lv * UPT + u];
    for (int q = t + NT * UPT; q < nU; q += NT) {  // surplus cells of an unusually large tile
      int32_t c = ut_cells[u0 + q];
      for (int lv = 0; lv < LC; ++lv) lds[lv * nup + q] = (k0 + lv < nlev) ? sf[(int64_t)(k0 + lv) * nsrc + c] : 0.0;
    }
    __syncthreads();
    const int kn1 = k0 + LC;
    if (kn1 < nlev) {
#pragma unroll
      for (int lv = 0; lv < LC; ++lv)
#pragma unroll
        for (int u = 0; u < UPT; ++u)
          pf[lv * UPT + u] = (cell[u] >= 0 && kn1 + lv < nlev) ? sf[(int64_t)(kn1 + lv) * nsrc + cell[u]] : 0.0;
    }
    const int kn = min(LC, nlev - k0);
    for (int kk = 0; kk < kn; ++kk) {
      const double *row = lds + kk * nup;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        double a = row[pts.l[r][0]], b = row[pts.l[r][1]], e = row[pts.l[r][2]];
        double val = wsum3(pts.ww[r][0], a, pts.ww[r][1], b, pts.ww[r][2], e);
        if (pts.act[r]) __builtin_nontemporal_store(pts.mapped[r] ? val : 0.0, df + (int64_t)(k0 + kk) * P + pts.off[r]);
      }
    }
    __syncthreads();
  }
}

typedef void (*lfu_fn)(const int32_t *, const int32_t *, const uint16_t *, const double *, const double *, double *, int, int, int, int64_t,
                       int, int, int, int, int);
struct LfuVariant { int txu, rpt, lc; lfu_fn fn; int nt = LFU_THREADS; };   // tile = txu x (nt * rpt / txu) points
static const LfuVariant g_cfu_variants[] = {  // cell-fast staged: a3_variant 100 + index
    {64, 2, 4, k_apply3_cfu_p<64, 2, 4, 8>},   {64, 2, 4, k_apply3_cfu_p<64, 2, 4, 12>},  {64, 2, 8, k_apply3_cfu_p<64, 2, 8, 16>},
    {64, 1, 4, k_apply3_cfu_p<64, 1, 4, 4>},   {64, 1, 8, k_apply3_cfu_p<64, 1, 8, 8>},   {64, 1, 8, k_apply3_cfu_p<64, 1, 8, 16>},
    {32, 2, 4, k_apply3_cfu_p<32, 2, 4, 8>},   {32, 1, 8, k_apply3_cfu_p<32, 1, 8, 8>},   {64, 2, 2, k_apply3_cfu_p<64, 2, 2, 4>},
    {64, 2, 16, k_apply3_cfu_p<64, 2, 16, 32>},
    // 10-12: 64 x 16-point tiles (smaller one-cell ring per point, more registers)
    {64, 4, 4, k_apply3_cfu_p<64, 4, 4, 16>},  {64, 4, 2, k_apply3_cfu_p<64, 4, 2, 8>},   {64, 4, 4, k_apply3_cfu_p<64, 4, 4, 20>},
    // 13: 64 x 8 tiles with room for 1024 unique cells per tile (C4 needs 756 of the 768 that variant 1 holds)
    {64, 2, 4, k_apply3_cfu_p<64, 2, 4, 16>},
    // 14-15: 64 x 32-point tiles served by 512 threads (4 points each): half the tile-edge re-reads of 64 x 16 where a tile
    // has few cells per point (a global lat-lon grid finer than its mesh, C5); room for 1536 / 2048 cells per tile
    {64, 4, 4, k_apply3_cfu_p<64, 4, 4, 12, 512>, 512}, {64, 4, 4, k_apply3_cfu_p<64, 4, 4, 16, 512>, 512},
    // 16-17: 64 x 16-point tiles on 512 threads (2 points each, the registers of the 64 x 8 kernel)
    {64, 2, 4, k_apply3_cfu_p<64, 2, 4, 16, 512>, 512}, {64, 2, 4, k_apply3_cfu_p<64, 2, 4, 12, 512>, 512},
    // 18: 64 x 32-point tiles on 1024 threads (2 points each)
    {64, 2, 4, k_apply3_cfu_p<64, 2, 4, 8, 1024>, 1024},
    // (compact tiles of 32 x 32, 16 x 64 and 32 x 16 points with the same 1024-cell capacity were measured in round 2 on C4,
    //  Morton-numbered C4, C2 and C5: 0-15 % slower than 64 x 8 / 64 x 16 everywhere, profiles/r02_sweep_cfu_compact.txt)
};
static int g_tile_band = 0;  // "tile_band": tile rows per band of the tile order (geom.h band_order); 0 = row-major
void mpg_set_tile_band(int v) { g_tile_band = v < 0 ? 0 : (v > 255 ? 255 : v); }
int mpg_tile_band() { return g_tile_band; }
static int g_cfu_fpw = 1;   // "cfu_fields_per_wg": fields of a bundle served by one workgroup of the staged cell-fast kernel
void mpg_cfu_set_fields_per_wg(int v) { g_cfu_fpw = v < 1 ? 1 : (v > 255 ? 255 : v); }
int mpg_cfu_num_variants() { return (int)(sizeof(g_cfu_variants) / sizeof(g_cfu_variants[0])); }
// unique cells per tile a variant keeps in registers (NPF / LC * 256); beyond it a slow synchronous path takes over
static const int g_cfu_npf[] = {8, 12, 16, 4, 8, 16, 8, 8, 4, 32, 16, 8, 20, 16, 12, 16, 16, 12, 8};
static int cfu_capacity(int variant) { return g_cfu_npf[variant] / g_cfu_variants[variant].lc * g_cfu_variants[variant].nt; }
static int lfu_build_shape(mpg_handle_s *h, int txu, int tyu, hipStream_t s, int cap = 1024, int dmax = 8);
static int cfu_build(mpg_handle_s *h, int variant, hipStream_t s) {
  const LfuVariant &v = g_cfu_variants[variant];
  return lfu_build_shape(h, v.txu, v.nt * v.rpt / v.txu, s, cfu_capacity(variant) > 1024 ? cfu_capacity(variant) : 1024, 16);
}
static const LfuVariant g_lfu_variants[] = {
    // 0-5: two-phase, 64-wide tiles
    {64, 1, 8, k_apply3_lfu<64, 1, 8>},   {64, 1, 16, k_apply3_lfu<64, 1, 16>}, {64, 2, 8, k_apply3_lfu<64, 2, 8>},
    {64, 2, 16, k_apply3_lfu<64, 2, 16>}, {64, 1, 4, k_apply3_lfu<64, 1, 4>},   {64, 2, 4, k_apply3_lfu<64, 2, 4>},
    // 6-11: software-pipelined, 64-wide tiles
    {64, 1, 8, k_apply3_lfu_p<64, 1, 8, 8>},    {64, 1, 16, k_apply3_lfu_p<64, 1, 16, 12>}, {64, 2, 8, k_apply3_lfu_p<64, 2, 8, 16>},
    {64, 2, 16, k_apply3_lfu_p<64, 2, 16, 16>}, {64, 1, 8, k_apply3_lfu_p<64, 1, 8, 12>},   {64, 1, 16, k_apply3_lfu_p<64, 1, 16, 16>},
    // 12-19: software-pipelined, compact tiles (32 x 8, 16 x 16, 32 x 16, 16 x 32 points)
    {32, 1, 16, k_apply3_lfu_p<32, 1, 16, 16>}, {16, 1, 16, k_apply3_lfu_p<16, 1, 16, 16>}, {32, 1, 8, k_apply3_lfu_p<32, 1, 8, 8>},
    {16, 1, 8, k_apply3_lfu_p<16, 1, 8, 8>},    {32, 2, 16, k_apply3_lfu_p<32, 2, 16, 16>}, {16, 2, 16, k_apply3_lfu_p<16, 2, 16, 16>},
    {32, 1, 16, k_apply3_lfu_p<32, 1, 16, 12>}, {32, 1, 32, k_apply3_lfu_p<32, 1, 32, 16>},
};
int mpg_lfu_num_variants() { return (int)(sizeof(g_lfu_variants) / sizeof(g_lfu_variants[0])); }

static int lfu_build(mpg_handle_s *h, int txu, int rpt, hipStream_t s) { return lfu_build_shape(h, txu, LFU_THREADS * rpt / txu, s); }


// tile lists for tiles of txu x tyu target points (cached in the handle, keyed by the shape)
static int lfu_build_shape(mpg_handle_s *h, int txu, int tyu, hipStream_t s, int cap, int dmax) {
  const int key = (txu * 1024 + tyu) | (dmax > 8 ? 1 << 24 : 0);   // the same shape under the two alignment rules: two sets of lists
  if (h->ut_rpt == key) return MPG_SUCCESS;
  if (h->ut2_rpt == key) {  // the other layout's shape: swap the parked lists in, no device work
    std::swap(h->ut_ptr, h->ut2_ptr);
    std::swap(h->ut_cells, h->ut2_cells);
    std::swap(h->lidx, h->lidx2);
    std::swap(h->ut_rpt, h->ut2_rpt);
    std::swap(h->ut_max, h->ut2_max);
    std::swap(h->ut_align, h->ut2_align);
    std::swap(h->ut_total, h->ut2_total);
    std::swap(h->ut_lines, h->ut2_lines);
    return MPG_SUCCESS;
  }
  int rc;
  if (h->ut_rpt) {  // park the lists in use (dropping what was parked) and build the new shape beside them
    h->ut2_ptr.free();
    h->ut2_cells.free();
    h->lidx2.free();
    h->ut2_ptr = h->ut_ptr;
    h->ut2_cells = h->ut_cells;
    h->lidx2 = h->lidx;
    h->ut2_rpt = h->ut_rpt;
    h->ut2_max = h->ut_max;
    h->ut2_align = h->ut_align;
    h->ut2_total = h->ut_total;
    h->ut2_lines = h->ut_lines;
    h->ut_ptr = DevBuf<int32_t>();
    h->ut_cells = DevBuf<int32_t>();
    h->lidx = DevBuf<uint16_t>();
  }
  h->ut_ptr.free();
  h->ut_cells.free();
  h->ut_rpt = 0;
  const bool big = 3 * txu * tyu > LFU_SORT;   // tiles of more than 1365 points sort in an 8192-entry buffer
  if (3 * txu * tyu > 2 * LFU_SORT) {
    mpg_set_error("staged Regrid: tile of %d x %d points exceeds the sort buffer", txu, tyu);
    return MPG_ERR_UNSUPPORTED;
  }
  // Row-shifted tiles (aligned store segments, mpg_internal.h) first; when their longest list does not fit the staged
  // kernels' 1024 cells per tile (grids whose rows start at many different offsets in a line: the shifted rows of a tile
  // then spread over up to 31 more columns) the lists are built for unshifted tiles instead.
  h->ut_align = mpg_tile_align(h->nx_dst, dmax);
  if (h->nx_dst % h->ut_align == 0) h->ut_align = 1;   // every row starts aligned already: nothing to shift
  const int nty = (h->ny_dst + tyu - 1) / tyu;
  int ntx = 0;
  int64_t ntile = 0;
  TmpBuf<int32_t> count;
  TmpBuf<char> tmp;
  TmpBuf<unsigned long long> scal;   // [0] distinct 128-byte lines, [1] (as int32) longest list
  int32_t tot32 = 0;
  for (;;) {
    ntx = mpg_tile_ntx(h->nx_dst, txu, h->ut_align);
    ntile = (int64_t)ntx * nty;
    count.free();
    h->ut_ptr.free();
    tmp.free();
    scal.free();
    if ((rc = count.alloc(ntile + 1)) || (rc = h->ut_ptr.alloc(ntile + 1))) return rc;
    if (!h->lidx.p && (rc = h->lidx.alloc(3 * (size_t)h->n_dst))) return rc;
    MPG_HIP(hipMemsetAsync(count.p, 0, sizeof(int32_t) * (ntile + 1), s));
    if (big)
      k_lfu_build<false, 2 * LFU_SORT><<<(unsigned)ntile, LFU_THREADS, 0, s>>>(h->idx.p, h->nx_dst, h->ny_dst, h->ut_align, txu, tyu, ntx, count.p,
                                                                              nullptr, nullptr, nullptr, nullptr);
    else
      k_lfu_build<false><<<(unsigned)ntile, LFU_THREADS, 0, s>>>(h->idx.p, h->nx_dst, h->ny_dst, h->ut_align, txu, tyu, ntx, count.p, nullptr, nullptr,
                                                                nullptr, nullptr);
    MPG_HIP(hipGetLastError());
    // list offsets and the longest list on the device (rocPRIM scan / reduce); only three scalars come back to the host
    size_t b_scan = 0, b_max = 0;
    MPG_HIP(rocprim::exclusive_scan(nullptr, b_scan, count.p, h->ut_ptr.p, (int32_t)0, (size_t)ntile + 1, rocprim::plus<int32_t>(), s));
    if ((rc = scal.alloc(2))) return rc;
    MPG_HIP(rocprim::reduce(nullptr, b_max, count.p, (int32_t *)(scal.p + 1), (int32_t)0, (size_t)ntile, rocprim::maximum<int32_t>(), s));
    if ((rc = tmp.alloc((b_scan > b_max ? b_scan : b_max) + 16))) return rc;
    MPG_HIP(hipMemsetAsync(scal.p, 0, 2 * sizeof(unsigned long long), s));
    MPG_HIP(rocprim::exclusive_scan((void *)tmp.p, b_scan, count.p, h->ut_ptr.p, (int32_t)0, (size_t)ntile + 1, rocprim::plus<int32_t>(), s));
    MPG_HIP(rocprim::reduce((void *)tmp.p, b_max, count.p, (int32_t *)(scal.p + 1), (int32_t)0, (size_t)ntile, rocprim::maximum<int32_t>(), s));
    int32_t max32 = 0;
    MPG_HIP(hipMemcpyAsync(&tot32, h->ut_ptr.p + ntile, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MPG_HIP(hipMemcpyAsync(&max32, (int32_t *)(scal.p + 1), sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MPG_HIP(hipStreamSynchronize(s));
    if (h->ut_align > 1 && max32 > cap) {
      h->ut_align = 1;
      continue;
    }
    break;
  }
  if (tot32 < 0) {   // the int32 scan wrapped
    mpg_set_error("tile cell lists exceed 2^31 entries");
    return MPG_ERR_OVERFLOW;
  }
  const int64_t tot = tot32;
  if ((rc = h->ut_cells.alloc((size_t)tot + 1))) return rc;
  if (big)
    k_lfu_build<true, 2 * LFU_SORT><<<(unsigned)ntile, LFU_THREADS, 0, s>>>(h->idx.p, h->nx_dst, h->ny_dst, h->ut_align, txu, tyu, ntx, nullptr,
                                                                           h->ut_ptr.p, h->ut_cells.p, h->lidx.p, scal.p);
  else
    k_lfu_build<true><<<(unsigned)ntile, LFU_THREADS, 0, s>>>(h->idx.p, h->nx_dst, h->ny_dst, h->ut_align, txu, tyu, ntx, nullptr, h->ut_ptr.p,
                                                             h->ut_cells.p, h->lidx.p, scal.p);
  MPG_HIP(hipGetLastError());
  unsigned long long hs[2] = {0, 0};
  MPG_HIP(hipMemcpyAsync(hs, scal.p, sizeof(hs), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  h->ut_rpt = key;
  h->ut_max = (int)(int32_t)(hs[1] & 0xffffffffu);
  h->ut_total = tot;
  h->ut_lines = (int64_t)hs[0];
  return MPG_SUCCESS;
}

int mpg_lfu_build_shape(mpg_handle_s *h, int txu, int tyu, hipStream_t s) { return lfu_build_shape(h, txu, tyu, s); }

// Which level-fast kernel serves this handle?  Measured on MI355X (profiles/r01_sweep_lfu.txt), 4 fields x 55 levels:
//   target points per source cell   row-gather k_apply3_lf   LDS-staged
//   1.4  (C4, 3 M cells)            4.76 TB/s                3.3 TB/s
//   2.9  (C2, 655 k cells)          2.61 TB/s                4.5 TB/s
//   2.5  (C5, global lat-lon)       2.80 TB/s                4.7 TB/s
// Staging pays when a staged row is referenced often enough; the statistic that separates the cases is
// reuse = 3 * n_dst / sum(unique cells per tile): ~2.5 on C4, 5-6 on C2 / C5.
static int g_lfu_auto_variant = 11;
#define LFU_AUTO_MIN_REUSE 3.5f
static float g_lfu_min_reuse = LFU_AUTO_MIN_REUSE;  // "lfu_min_reuse_x10" knob (level-fast choice only; decided at a handle's first call)
void mpg_lfu_set_min_reuse_x10(int v) { g_lfu_min_reuse = 0.1f * (float)v; }
void mpg_lfu_set_auto_variant(int v) { g_lfu_auto_variant = v; }
int mpg_lfu_auto(mpg_handle_s *h, hipStream_t s, int *lfu_variant) {
  const LfuVariant &v = g_lfu_variants[g_lfu_auto_variant];
  if (h->lf_choice == 0) {
    int rc = lfu_build(h, v.txu, v.rpt, s);
    if (rc) return rc;
    h->lf_reuse = h->ut_total > 0 ? 3.0f * (float)h->n_dst / (float)h->ut_total : 0.f;
    h->lf_choice = h->lf_reuse >= g_lfu_min_reuse ? 1 : -1;
    if (h->lf_choice < 0 && h->cf_choice <= 0) {  // not needed: give the memory back
      h->ut_ptr.free();
      h->ut_cells.free();
      h->lidx.free();
      h->ut_rpt = 0;
    }
  }
  *lfu_variant = h->lf_choice > 0 ? g_lfu_auto_variant : -1;
  return MPG_SUCCESS;
}

// cell-fast: same statistic (measured: C2 reuse 5+ -> staged 1.5x faster; C4 reuse 2.5 -> k_apply3_cf is at the HBM limit)
// Per-handle choice (a3_staged = -1, the default), measured on MI355X (profiles/r01_sweep_cfu.txt, clean re-run):
//   tiles of 64 x 16 points (variant 10) when cells are shared a lot (reuse >= 3.5) and the lists fit: C2 0.82 ms,
//     C5 3.07 ms per 4 fields (64 x 8 tiles: 0.87 / 3.30; lane-gather 1.50 / 4.91);
//   else tiles of 64 x 8 points (variant 13: room for 1024 cells per tile, C4 needs 756) when the lists fit: C4 1.57 ms
//     (lane-gather 1.61);
//   else the lane-gather kernel (fine mesh under a coarse grid).
// cf_choice holds variant + 1, or -1 for the lane-gather kernel.
#define CFU_WIDE 10
#define CFU_BASE 13
#define CFU_TALL 16
// Round 2: when cells are not shared enough for the 256-thread 64 x 16 kernel, the same 64 x 16 tiles on 512 threads
// (variant 16: two points per thread like the 64 x 8 kernel, 2048 cells per tile) serve the handle: equal to 64 x 8 on the
// row-numbered C4 (4.91 ms both), 5 % faster on the Morton-numbered one (4.72 vs 4.96 ms), and the lists built for the
// reuse statistic are the ones used (one list build per handle instead of two).
int mpg_cfu_auto(mpg_handle_s *h, hipStream_t s, int *cfu_variant) {
  if (h->cf_choice == 0 || h->cf_for != -1) {
    h->cf_for = -1;
    int rc = cfu_build(h, CFU_TALL, s);   // 64 x 16 lists, shifted rows as long as 2048 cells per tile hold them
    if (rc) return rc;
    float reuse = h->ut_total > 0 ? 3.0f * (float)h->n_dst / (float)h->ut_total : 0.f;
    if (reuse >= LFU_AUTO_MIN_REUSE && h->ut_max <= cfu_capacity(CFU_WIDE)) {
      h->cf_choice = CFU_WIDE + 1;
    } else if (h->ut_max <= cfu_capacity(CFU_TALL)) {
      h->cf_choice = CFU_TALL + 1;
    } else {
      if ((rc = cfu_build(h, CFU_BASE, s))) return rc;
      h->cf_choice = h->ut_max <= cfu_capacity(CFU_BASE) ? CFU_BASE + 1 : -1;
    }
    if (h->cf_choice < 0 && h->lf_choice <= 0) {
      h->ut_ptr.free();
      h->ut_cells.free();
      h->lidx.free();
      h->ut_rpt = 0;
    }
  }
  *cfu_variant = h->cf_choice > 0 ? h->cf_choice - 1 : -1;
  return MPG_SUCCESS;
}

// Typed form of the staged cell-fast kernel (mpg_regrid_typed_dev): float32 or float64 source as the MPAS file stores
// it, float64 arithmetic (wsum3), dst = (TD)(value * scale + offset) -- the writer's T - 300 / PHB * 9.81 / NF90_FLOAT
// conversion fused in.  Fixed shape <64 x 8 points, 4 levels per chunk, 16 prefetch registers> = the f64 base variant.
template <typename TS, typename TD, int RPT = 2, int NT = LFU_THREADS>
__global__ __launch_bounds__(NT) void k_apply3_cfu_t(const int32_t *__restrict__ ut_ptr, const int32_t *__restrict__ ut_cells,
                                                              const uint16_t *__restrict__ lidx, const double *__restrict__ w,
                                                              const TS *__restrict__ src, TD *__restrict__ dst, int nx, int ny, int talign, int64_t nsrc,
                                                              int nlev, int ntx, int nty, int nfields, int ut_max, double scale,
                                                              double offset) {
  constexpr int TXU = 64, LC = 4, NPF = 16, UPT = NPF / LC;
  extern __shared__ double lds[];  // [LC][nup]
  const int nup = ut_max;
  const int64_t P = (int64_t)nx * ny;
  const unsigned ntile = (unsigned)ntx * nty;
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned tile = lin % ntile;
  const int f = lin / ntile;
  const int t = threadIdx.x;
  const int u0 = ut_ptr[tile], nU = ut_ptr[tile + 1] - u0;
  LfuPoints<TXU, RPT, NT> pts;
  pts.load(lidx, w, nx, ny, talign, tile % ntx, tile / ntx, 1);
  const TS *sf = src + (int64_t)f * nlev * nsrc;
  TD *df = dst + (int64_t)f * nlev * P;
  int32_t cell[UPT];
#pragma unroll
  for (int u = 0; u < UPT; ++u) {
    int q = t + NT * u;
    cell[u] = q < nU ? ut_cells[u0 + q] : -1;
  }
  TS pf[NPF];
#pragma unroll
  for (int lv = 0; lv < LC; ++lv)
#pragma unroll
    for (int u = 0; u < UPT; ++u) pf[lv * UPT + u] = (cell[u] >= 0 && lv < nlev) ? sf[(int64_t)lv * nsrc + cell[u]] : (TS)0;
  for (int k0 = 0; k0 < nlev; k0 += LC) {
#pragma unroll
    for (int lv = 0; lv < LC; ++lv)
#pragma unroll
      for (int u = 0; u < UPT; ++u)
        if (cell[u] >= 0) lds[lv * nup + t + NT * u] = (double)pf[lv * UPT + u];
    for (int q = t + NT * UPT; q < nU; q += NT) {
      int32_t c = ut_cells[u0 + q];
      for (int lv = 0; lv < LC; ++lv) lds[lv * nup + q] = (k0 + lv < nlev) ? (double)sf[(int64_t)(k0 + lv) * nsrc + c] : 0.0;
    }
    __syncthreads();
    const int kn1 = k0 + LC;
    if (kn1 < nlev) {
#pragma unroll
      for (int lv = 0; lv < LC; ++lv)
#pragma unroll
        for (int u = 0; u < UPT; ++u)
          pf[lv * UPT + u] = (cell[u] >= 0 && kn1 + lv < nlev) ? sf[(int64_t)(kn1 + lv) * nsrc + cell[u]] : (TS)0;
    }
    const int kn = min(LC, nlev - k0);
    for (int kk = 0; kk < kn; ++kk) {
      const double *row = lds + kk * nup;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        double a = row[pts.l[r][0]], b = row[pts.l[r][1]], e = row[pts.l[r][2]];
        double val = pts.mapped[r] ? wsum3(pts.ww[r][0], a, pts.ww[r][1], b, pts.ww[r][2], e) : 0.0;
        if (pts.act[r]) __builtin_nontemporal_store((TD)fma(val, scale, offset), df + (int64_t)(k0 + kk) * P + pts.off[r]);
      }
    }
    __syncthreads();
  }
}

template <typename TS, typename TD, int RPT, int NT>
static int launch_cfu_t(mpg_handle_s *h, const void *src, int nlev, int nfields, void *dst, double scale, double offset, hipStream_t s) {
  constexpr int TYU = NT * RPT / 64;
  const int ntx = mpg_tile_ntx(h->nx_dst, 64, h->ut_align), nty = (h->ny_dst + TYU - 1) / TYU;
  const size_t um = h->ut_max > 0 ? h->ut_max : 1;
  size_t lds = sizeof(double) * um * 4 + 16;
  if (lds > 160 * 1024) {
    mpg_set_error("Regrid(CELL_FAST, staged): %d unique cells per tile exceed the LDS", h->ut_max);
    return MPG_ERR_UNSUPPORTED;
  }
  auto fn = k_apply3_cfu_t<TS, TD, RPT, NT>;
  if (lds > 48 * 1024) MPG_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  fn<<<(unsigned)ntx * nty * nfields, NT, lds, s>>>(h->ut_ptr.p, h->ut_cells.p, h->lidx.p, h->w.p, (const TS *)src, (TD *)dst, h->nx_dst, h->ny_dst,
                                                   h->ut_align, h->n_src, nlev, ntx, nty, nfields, (int)um, scale, offset);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

// Typed form of the staged LEVEL-fast kernel: float32 rows as the MPAS history file stores them ([nCells][nVertLevels],
// 220 bytes per cell at 55 levels) or float64 rows, float64 arithmetic, dst = (TD)(value * scale + offset).  This is the
// path of a file-order, single-precision ingest with single-precision output.  Fixed shape <64 x 4 points, 16 levels per
// chunk, 16 prefetch registers> = the float64 auto variant.
template <typename TS, typename TD>
__global__ __launch_bounds__(LFU_THREADS) void k_apply3_lfu_t(const int32_t *__restrict__ ut_ptr, const int32_t *__restrict__ ut_cells,
                                                              const uint16_t *__restrict__ lidx, const double *__restrict__ w,
                                                              const TS *__restrict__ src, TD *__restrict__ dst, int nx, int ny, int talign, int64_t nsrc,
                                                              int nlev, int ntx, int nty, int nfields, int ut_max, double scale,
                                                              double offset) {
  constexpr int TXU = 64, RPT = 1, LC = 16, NPF = 16, LS = LC + 1, RPP = LFU_THREADS / LC;
  extern __shared__ double lds[];
  int32_t *cells = (int32_t *)(lds + (size_t)ut_max * LS);
  const int64_t P = (int64_t)nx * ny;
  const unsigned ntile = (unsigned)ntx * nty;
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned tile = lin % ntile;
  const int f = lin / ntile;
  const int t = threadIdx.x;
  const int u0 = ut_ptr[tile], nU = ut_ptr[tile + 1] - u0;
  for (int r = t; r < nU; r += LFU_THREADS) cells[r] = ut_cells[u0 + r];
  LfuPoints<TXU, RPT> pts;
  pts.load(lidx, w, nx, ny, talign, tile % ntx, tile / ntx, LS);
  const TS *sf = src + (int64_t)f * nlev * nsrc;
  TD *df = dst + (int64_t)f * nlev * P;
  const int lrow = t / LC, llev = t % LC;
  __syncthreads();
  int64_t roff[NPF];
#pragma unroll
  for (int u = 0; u < NPF; ++u) {
    int row = lrow + u * RPP;
    roff[u] = row < nU ? (int64_t)cells[row] * nlev + llev : -1;
  }
  TS pf[NPF];
#pragma unroll
  for (int u = 0; u < NPF; ++u) pf[u] = (roff[u] >= 0 && llev < nlev) ? sf[roff[u]] : (TS)0;
  for (int k0 = 0; k0 < nlev; k0 += LC) {
#pragma unroll
    for (int u = 0; u < NPF; ++u)
      if (roff[u] >= 0) lds[(lrow + u * RPP) * LS + llev] = (double)pf[u];
    for (int row = lrow + NPF * RPP; row < nU; row += RPP)
      lds[row * LS + llev] = (k0 + llev < nlev) ? (double)sf[(int64_t)cells[row] * nlev + k0 + llev] : 0.0;
    __syncthreads();
    const int kn1 = k0 + LC;
    if (kn1 < nlev) {
      const bool ok = kn1 + llev < nlev;
#pragma unroll
      for (int u = 0; u < NPF; ++u) pf[u] = (roff[u] >= 0 && ok) ? sf[roff[u] + kn1] : (TS)0;
    }
    const int kn = min(LC, nlev - k0);
    for (int kk = 0; kk < kn; ++kk) {
      double a = lds[pts.l[0][0] + kk], b = lds[pts.l[0][1] + kk], e = lds[pts.l[0][2] + kk];
      double val = pts.mapped[0] ? wsum3(pts.ww[0][0], a, pts.ww[0][1], b, pts.ww[0][2], e) : 0.0;
      if (pts.act[0]) __builtin_nontemporal_store((TD)fma(val, scale, offset), df + (int64_t)(k0 + kk) * P + pts.off[0]);
    }
    __syncthreads();
  }
}

template <typename TS, typename TD>
static int launch_lfu_t(mpg_handle_s *h, const void *src, int nlev, int nfields, void *dst, double scale, double offset, hipStream_t s) {
  const int ntx = mpg_tile_ntx(h->nx_dst, 64, h->ut_align), nty = (h->ny_dst + 3) / 4;
  const size_t um = h->ut_max > 0 ? h->ut_max : 1;
  size_t lds = sizeof(double) * um * 17 + sizeof(int32_t) * um + 16;
  if (lds > 160 * 1024) return MPG_ERR_UNSUPPORTED;
  if (lds > 48 * 1024)
    MPG_HIP(hipFuncSetAttribute((const void *)k_apply3_lfu_t<TS, TD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  k_apply3_lfu_t<TS, TD><<<(unsigned)ntx * nty * nfields, LFU_THREADS, lds, s>>>(h->ut_ptr.p, h->ut_cells.p, h->lidx.p, h->w.p, (const TS *)src,
                                                                                (TD *)dst, h->nx_dst, h->ny_dst, h->ut_align, h->n_src, nlev, ntx, nty,
                                                                                nfields, (int)um, scale, offset);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

// ---- level-fast, whole rows resident ("lfr") -------------------------------------------------------------------
// For handles with little sharing between target points (C4: 1.4 points per cell) chunking the levels costs more than
// it saves (every 128-byte line is touched by two chunks).  Here a tile of 32 x 4 target points keeps the COMPLETE
// rows of its unique cells in LDS: each row is fetched once, as one contiguous wave-wide load (lanes = levels), then
// 256 threads combine: thread = (point, level parity), so a wave still stores 64 consecutive points of one level
// (two 256-byte row segments).  LDS holds the source element type (float32 rows stay float32; widened when read).
template <typename TS, typename TD>
__global__ __launch_bounds__(LFU_THREADS) void k_apply3_lfr(const int32_t *__restrict__ ut_ptr, const int32_t *__restrict__ ut_cells,
                                                            const uint16_t *__restrict__ lidx, const double *__restrict__ w,
                                                            const TS *__restrict__ src, TD *__restrict__ dst, int nx, int ny, int talign, int64_t nsrc,
                                                            int nlev, int nlp, int ntx, int nty, int nfields, double scale, double offset) {
  constexpr int TXU = 32, TYU = 4, NP = TXU * TYU;
  extern __shared__ double lds_raw[];
  TS *rows = (TS *)lds_raw;  // [nU][nlp]
  const int64_t P = (int64_t)nx * ny;
  const unsigned ntile = (unsigned)ntx * nty;
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned tile = lin % ntile;
  const int f = lin / ntile;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int u0 = ut_ptr[tile], nU = ut_ptr[tile + 1] - u0;
  const TS *sf = src + (int64_t)f * nlev * nsrc;
  // phase 1: wave w fetches rows w, w+4, ...: 4 rows in flight per wave and iteration
  for (int rb = wave; rb < nU; rb += 16) {
    TS v[4];
    int c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int row = rb + 4 * u;
      c[u] = row < nU ? ut_cells[u0 + row] : -1;
    }
    for (int l0 = 0; l0 < nlev; l0 += 64) {
      const bool ok = l0 + lane < nlev;
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (c[u] >= 0 && ok) ? sf[(int64_t)c[u] * nlev + l0 + lane] : (TS)0;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c[u] >= 0 && ok) rows[(rb + 4 * u) * nlp + l0 + lane] = v[u];
    }
  }
  // this thread's point
  const int pt = t % NP, par = t / NP;
  const int j = (tile / ntx) * TYU + pt / TXU, i = (tile % ntx) * TXU + pt % TXU - mpg_tile_shift(j, nx, talign);
  const bool act = i >= 0 && i < nx && j < ny;
  const int64_t p = act ? (int64_t)j * nx + i : 0;
  int l0 = lidx[p], l1 = lidx[P + p], l2 = lidx[2 * P + p];
  const double w0 = w[p], w1 = w[P + p], w2 = w[2 * P + p];
  const bool mapped = l0 != 0xFFFF;
  l0 = mapped ? l0 * nlp : 0;
  l1 = mapped ? l1 * nlp : 0;
  l2 = mapped ? l2 * nlp : 0;
  __syncthreads();
  TD *df = dst + (int64_t)f * nlev * P + p;
  if (act)
    for (int k = par; k < nlev; k += 2) {
      double val = mapped ? wsum3(w0, (double)rows[l0 + k], w1, (double)rows[l1 + k], w2, (double)rows[l2 + k]) : 0.0;
      __builtin_nontemporal_store((TD)fma(val, scale, offset), df + (int64_t)k * P);
    }
}

template <typename TS, typename TD>
static int launch_lfr(mpg_handle_s *h, const void *src, int nlev, int nfields, void *dst, double scale, double offset, hipStream_t s) {
  const int ntx = mpg_tile_ntx(h->nx_dst, 32, h->ut_align), nty = (h->ny_dst + 3) / 4;
  const int nlp = nlev | 1;  // odd row stride: conflict-free column reads
  const size_t um = h->ut_max > 0 ? h->ut_max : 1;
  size_t lds = sizeof(TS) * um * nlp + 16;
  if (lds > 80 * 1024) return MPG_ERR_UNSUPPORTED;  // fewer than two workgroups per CU: not worth it
  if (lds > 48 * 1024) MPG_HIP(hipFuncSetAttribute((const void *)k_apply3_lfr<TS, TD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  k_apply3_lfr<TS, TD><<<(unsigned)ntx * nty * nfields, LFU_THREADS, lds, s>>>(h->ut_ptr.p, h->ut_cells.p, h->lidx.p, h->w.p, (const TS *)src,
                                                                              (TD *)dst, h->nx_dst, h->ny_dst, h->ut_align, h->n_src, nlev, nlp, ntx, nty, nfields,
                                                                              scale, offset);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

// explicit entry (lf_variant 200): rows-resident kernel for any element types
int mpg_k_apply3_lfr(mpg_handle_s *h, const void *src, int src_f32, int nlev, int nfields, void *dst, int dst_f32, double scale, double offset,
                     hipStream_t s) {
  int rc = lfu_build_shape(h, 32, 4, s);
  if (rc) return rc;
  if (src_f32 && dst_f32) return launch_lfr<float, float>(h, src, nlev, nfields, dst, scale, offset, s);
  if (src_f32) return launch_lfr<float, double>(h, src, nlev, nfields, dst, scale, offset, s);
  if (dst_f32) return launch_lfr<double, float>(h, src, nlev, nfields, dst, scale, offset, s);
  return launch_lfr<double, double>(h, src, nlev, nfields, dst, scale, offset, s);
}

// -> MPG_ERR_UNSUPPORTED when the row-gather kernel is the better choice for this handle (caller falls back)
int mpg_k_apply3_lfu_typed(mpg_handle_s *h, const void *src, int src_f32, int nlev, int nfields, void *dst, int dst_f32, double scale,
                           double offset, hipStream_t s) {
  int pick, rc = mpg_lfu_auto(h, s, &pick);
  if (rc) return rc;
  if (pick < 0) return MPG_ERR_UNSUPPORTED;
  if ((rc = lfu_build(h, 64, 1, s))) return rc;
  if (src_f32 && dst_f32) return launch_lfu_t<float, float>(h, src, nlev, nfields, dst, scale, offset, s);
  if (src_f32) return launch_lfu_t<float, double>(h, src, nlev, nfields, dst, scale, offset, s);
  if (dst_f32) return launch_lfu_t<double, float>(h, src, nlev, nfields, dst, scale, offset, s);
  return launch_lfu_t<double, double>(h, src, nlev, nfields, dst, scale, offset, s);
}

template <int RPT, int NT>
static int launch_cfu_t_types(mpg_handle_s *h, const void *src, int src_f32, int nlev, int nfields, void *dst, int dst_f32, double scale,
                              double offset, hipStream_t s) {
  if (src_f32 && dst_f32) return launch_cfu_t<float, float, RPT, NT>(h, src, nlev, nfields, dst, scale, offset, s);
  if (src_f32) return launch_cfu_t<float, double, RPT, NT>(h, src, nlev, nfields, dst, scale, offset, s);
  if (dst_f32) return launch_cfu_t<double, float, RPT, NT>(h, src, nlev, nfields, dst, scale, offset, s);
  return launch_cfu_t<double, double, RPT, NT>(h, src, nlev, nfields, dst, scale, offset, s);
}

// The typed entry follows the handle's per-handle choice of tile shape (mpg_cfu_auto: 64 x 16 points on 256 or 512
// threads, else 64 x 8), so that float64 and typed Regrids of one handle share ONE set of tile lists.
int mpg_k_apply3_cfu_typed(mpg_handle_s *h, const void *src, int src_f32, int nlev, int nfields, void *dst, int dst_f32, double scale,
                           double offset, hipStream_t s) {
  int variant = -1, rc;
  if (mpg_a3_staged() == -1) {
    if ((rc = mpg_cfu_auto(h, s, &variant))) return rc;
    if (variant < 0) return MPG_ERR_UNSUPPORTED;   // caller falls back to the lane-gather typed kernel
  }
  if (variant == CFU_TALL || variant == CFU_WIDE) {
    if ((rc = cfu_build(h, variant, s))) return rc;
    if (variant == CFU_TALL) return launch_cfu_t_types<2, 512>(h, src, src_f32, nlev, nfields, dst, dst_f32, scale, offset, s);
    return launch_cfu_t_types<4, LFU_THREADS>(h, src, src_f32, nlev, nfields, dst, dst_f32, scale, offset, s);
  }
  if ((rc = cfu_build(h, CFU_BASE, s))) return rc;
  if (h->ut_max > 1024) return MPG_ERR_UNSUPPORTED;
  return launch_cfu_t_types<2, LFU_THREADS>(h, src, src_f32, nlev, nfields, dst, dst_f32, scale, offset, s);
}

// Does the explicit staged variant suit this handle?  Tiles whose points share almost no cells (a fine mesh under a coarse
// grid: up to 3 cells per point) overflow the register-resident part of the list; the lane-gather kernel is the right tool
// there.  The decision is cached in the handle (cf_choice) so the lists are built once.
int mpg_cfu_fits(mpg_handle_s *h, int variant, hipStream_t s, int *fits) {
  if (h->cf_choice == 0 || h->cf_for != variant) {
    h->cf_for = variant;
    int rc = cfu_build(h, variant, s);
    if (rc) return rc;
    h->cf_choice = h->ut_max <= cfu_capacity(variant) ? variant + 1 : -1;
    if (h->cf_choice < 0 && h->lf_choice <= 0) {
      h->ut_ptr.free();
      h->ut_cells.free();
      h->lidx.free();
      h->ut_rpt = 0;
    }
  }
  *fits = h->cf_choice > 0;
  return MPG_SUCCESS;
}

int mpg_k_apply3_cfu(mpg_handle_s *h, int variant, const double *src, int nlev, int nfields, double *dst, hipStream_t s) {
  const LfuVariant &v = g_cfu_variants[variant];
  int rc = cfu_build(h, variant, s);
  if (rc) return rc;
  const int tyu = v.nt * v.rpt / v.txu;
  const int ntx = mpg_tile_ntx(h->nx_dst, v.txu, h->ut_align), nty = (h->ny_dst + tyu - 1) / tyu;
  const size_t um = h->ut_max > 0 ? h->ut_max : 1;
  size_t lds = sizeof(double) * um * v.lc + 16;
  if (lds > 160 * 1024) {
    mpg_set_error("Regrid(CELL_FAST, staged): %d unique cells per tile exceed the LDS", h->ut_max);
    return MPG_ERR_UNSUPPORTED;
  }
  if (lds > 48 * 1024) MPG_HIP(hipFuncSetAttribute((const void *)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (nfields > 0xffff) {
    mpg_set_error("Regrid: more than 65535 fields in one bundle");
    return MPG_ERR_UNSUPPORTED;
  }
  const int fpw = g_cfu_fpw < nfields ? g_cfu_fpw : nfields;
  const int ngroups = (nfields + fpw - 1) / fpw;
  v.fn<<<(unsigned)ntx * nty * ngroups, v.nt, lds, s>>>(h->ut_ptr.p, h->ut_cells.p, h->lidx.p, h->w.p, src, dst, h->nx_dst, h->ny_dst, h->ut_align,
                                                              h->n_src, nlev, ntx, nty, nfields | ((fpw & 0xff) << 16) | (g_tile_band << 24), (int)um);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

int mpg_k_apply3_lfu(mpg_handle_s *h, int variant, const double *src, int nlev, int nfields, double *dst, hipStream_t s) {
  const LfuVariant &v = g_lfu_variants[variant];
  int rc = lfu_build(h, v.txu, v.rpt, s);
  if (rc) return rc;
  const int tyu = LFU_THREADS * v.rpt / v.txu;
  const int ntx = mpg_tile_ntx(h->nx_dst, v.txu, h->ut_align), nty = (h->ny_dst + tyu - 1) / tyu;
  const size_t um = h->ut_max > 0 ? h->ut_max : 1;  // unmapped points read row 0
  size_t lds = sizeof(double) * um * (v.lc + 1) + sizeof(int32_t) * um + 16;
  if (lds > 160 * 1024) {
    mpg_set_error("Regrid(LEV_FAST, staged): %d unique cells per tile exceed the LDS", h->ut_max);
    return MPG_ERR_UNSUPPORTED;
  }
  if (lds > 48 * 1024) MPG_HIP(hipFuncSetAttribute((const void *)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  v.fn<<<(unsigned)ntx * nty * nfields, LFU_THREADS, lds, s>>>(h->ut_ptr.p, h->ut_cells.p, h->lidx.p, h->w.p, src, dst, h->nx_dst, h->ny_dst, h->ut_align,
                                                              h->n_src, nlev, ntx, nty, nfields, (int)um);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}
