// LDS-staged 3-point Regrid: every tile of target points carries the sorted list of the source cells its points
// reference and each point keeps three 16-bit positions in that list; the kernels load each unique cell ONCE per tile
// (and level chunk), park the values in LDS and let every thread combine its points from there.
//
//   k_lfu_build       per-tile unique cell lists + ranks, built once per handle on the device (one pass: sort in LDS,
//                     unique, ranks; lists at a fixed stride so that no count / scan pass is needed first)
//   k_apply3_cfu      cell-fastest source [nlev][ncell] (the reference's in-memory order, input_data.F90:653-655): lanes
//                     along the sorted cell list (neighbouring ids, coalesced), 4 levels per chunk, chunk c+1 prefetched
//                     into registers while chunk c is combined; three tile shapes (g_cfu_variants)
//   k_apply3_lfu      level-fastest source [ncell][nlev] (MPAS file order, :630,645), 64 x 8-point tiles, 16 levels per
//                     chunk, lanes along the levels, the same register prefetch; branch-free body on buffer addressing
//
// Why staging: with 2.5-2.9 target points per source cell (BASELINE configs 2, 3, 5) the gather kernels fetch every
// value ~3x through L2 -> CU and end up latency / issue bound at 2.6-3.3 TB/s; staged, the same workloads run at the HBM
// rate (profiles/r01_sweep_cfu.txt, r01_sweep_lfu.txt).  With 1.4 points per cell (config 4) the cell-fast form is equal
// to the lane gather and the level-fast form loses to the row gather, so mpg_cfu_auto / mpg_lfu_auto choose per handle
// by  reuse = 3 * n_dst / sum(unique cells per tile).
// Arithmetic = wsum3 (geom.h): bit-identical to the gather kernels.  The typed forms (float32 / float64 on either side,
// big-endian on either side, dst = (TD)(value * scale + offset)) are the same templates: mpg_regrid_dev is <double,
// double> without the epilogue.
// Shapes measured in rounds 1-2 and dropped from the library in round 3 (two-phase form without prefetch, 32- and
// 16-wide tiles, 64 x 32 tiles on 512 / 1024 threads, 2 / 8 / 16 levels per chunk, several fields per workgroup, banded
// tile order, rows-resident 32 x 4 and 64 x 1 / 2 / 4 tiles with deep prefetch; round 3: rows-resident 64 x 8 tiles with a
// per-tile level chunk, two levels per lane in k_apply3_lfu, 128-byte level chunks with the slab in the source type,
// vertical tile bands -- all slower or equal on configuration 5): profiles/r01_sweep_cfu.txt, r01_sweep_lfu.txt, r02_sweep_cfu_compact.txt,
// r02_sweep_cfu_fpw.txt, r02_tile_order_and_height.txt, r02_lfs_*.txt, r03_lf_experiments.txt.
#include <limits.h>
#include <string.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "geom.h"
#include "mpg_internal.h"

int mpg_field_band(int kernel_default);
static int g_staged_lds_pad_kb = 0;   // "staged_lds_pad_kb" knob (A/B): extra dynamic LDS per workgroup of k_apply3_cfu / k_apply3_lfu
void mpg_set_staged_lds_pad_kb(int v) { g_staged_lds_pad_kb = v; }
int mpg_staged_lds_pad_kb() { return g_staged_lds_pad_kb; }
#define LFU_THREADS 256
#define LFU_LIST_PAD 1024  // a tile's list is padded with its last cell up to the row count of its class: 64, 128 ... this many (min stride)
#define LFU_SORT 4096   // sort buffer: 3 ids x (at most) 1024 points, padded to a power of two

// ---- per-tile unique cell lists (set-up, runtime tile shape) --------------------------------------------------
// One workgroup per tile of txu x tyu points: the 3*np cell ids are sorted in LDS (bitonic), duplicates dropped, and each
// point's three ids are replaced by their rank in the tile's list.  The list of tile t starts at ut_cells[t * stride]
// (stride = 3*np rounded up: the hard upper bound), its length goes to ut_cnt[t]; stats[0] += distinct groups of 16
// consecutive ids (lines of a cell-fast float64 field), stats[1] = max length, stats[2] += length.
// FILL = false only measures (stats) the tiles blockIdx.x * tile_step: the sampled reuse statistic of mpg_lfu_auto.
template <bool FILL, int SB = LFU_SORT>
__global__ __launch_bounds__(LFU_THREADS) void k_lfu_build(const int32_t *__restrict__ idx, int nx, int ny, int talign, int txu, int tyu, int ntx,
                                                           int tile_step, int32_t *__restrict__ ut_cnt, int32_t *__restrict__ ut_cells, int stride,
                                                           uint16_t *__restrict__ lidx, unsigned long long *__restrict__ stats) {
  constexpr int PER = SB / LFU_THREADS;
  __shared__ int32_t keys[SB];
  __shared__ int32_t part[LFU_THREADS + 1];
  const int np = txu * tyu, nk = 3 * np;
  const int64_t P = (int64_t)nx * ny;
  const int tile = blockIdx.x * tile_step, tx = tile % ntx, ty = tile / ntx, t = threadIdx.x;
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
  if (t == 0) {
    atomicMax(stats + 1, (unsigned long long)total);
    atomicAdd(stats + 2, (unsigned long long)total);
  }
  if (!FILL) return;
  if (t == 0) ut_cnt[tile] = total;
  const int base = part[t];   // every read of keys[] happened before the barriers above: compact in place
  int32_t *out = ut_cells + (int64_t)tile * stride;
  for (int q = 0; q < nm; ++q) {
    keys[base + q] = mine[q];
    out[base + q] = mine[q];
  }
  __syncthreads();
  // entries total .. LFU_LIST_PAD-1 repeat the last cell (cell 0 for an empty tile): k_apply3_lfu reads a fixed number of
  // entries per tile without looking at the count first
  {
    const int32_t last = total > 0 ? keys[total - 1] : 0;
    int cls_rows = 64;                       // the row count of the tile's class (launch_lfu: 64 << c)
    while (cls_rows < total && cls_rows < LFU_LIST_PAD) cls_rows <<= 1;
    const int pad_to = stride < cls_rows ? stride : cls_rows;
    for (int e = total + t; e < pad_to; e += LFU_THREADS) out[e] = last;
  }
  // locality statistic: distinct groups of 16 consecutive ids (= 128-byte lines of a cell-fast float64 field) in the list
  {
    int nl = 0;
    for (int e = t; e < total; e += LFU_THREADS) nl += e == 0 || (keys[e] >> 4) != (keys[e - 1] >> 4);
    for (int o = 32; o > 0; o >>= 1) nl += __shfl_down(nl, o);
    if ((t & 63) == 0 && nl) atomicAdd(stats, (unsigned long long)nl);
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
// Thread t serves points pt = t + NT*r (r < RPT) of the tile in row-major order: a wave covers 64 consecutive
// points = one row of 64 points (512-byte float64 / 256-byte float32 store segments, aligned by the row shift).
template <int RPT, int NT>
struct LfuPoints {
  int l[RPT][3];
  double ww[RPT][3];
  bool act[RPT], mapped[RPT];
  int64_t off[RPT];  // j*nx + i
  __device__ __forceinline__ void load(const uint16_t *__restrict__ lidx, const double *__restrict__ w, int nx, int ny, int talign, int tx, int ty, int LS) {
    constexpr int TY = NT * RPT / 64;
    const int64_t P = (int64_t)nx * ny;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      int pt = (int)threadIdx.x + NT * r;
      int j = ty * TY + pt / 64, i = tx * 64 + pt % 64 - mpg_tile_shift(j, nx, talign);
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

// cell-fastest source: per chunk of LC = 4 levels the workgroup loads the tile's unique cells once (lanes along the sorted
// cell list) into LDS [LC][nup]; chunk c+1 is prefetched into registers (UPT cells x LC levels per thread) while chunk c
// is combined and stored.  Tiles with more than UPT * NT unique cells load the surplus synchronously.
template <typename TS, typename TD, int RPT, int NT, bool EPI>
__global__ __launch_bounds__(NT) void k_apply3_cfu(const int32_t *__restrict__ ut_cnt, const int32_t *__restrict__ ut_cells, int stride,
                                                   const uint16_t *__restrict__ lidx, const double *__restrict__ w,
                                                   const TS *__restrict__ src, TD *__restrict__ dst, int nx, int ny, int talign, int64_t nsrc,
                                                   int nlev, int ntx, int nty, int ut_max, double scale, double offset, int band, int st_all_nt, FieldTab tab) {
  constexpr int LC = 4, UPT = 4, NPF = LC * UPT;
  extern __shared__ double lds[];  // [LC][nup]
  const int nup = ut_max;
  const int64_t P = (int64_t)nx * ny;
  const unsigned ntile = (unsigned)ntx * nty;
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  unsigned tile;
  int f;
  band_map(lin, ntile, gridDim.x / ntile, (unsigned)band, tile, f);
  const int t = threadIdx.x;
  const int nU = ut_cnt[tile];
  const int32_t *cells = ut_cells + (int64_t)tile * stride;
  LfuPoints<RPT, NT> pts;
  pts.load(lidx, w, nx, ny, talign, tile % ntx, tile / ntx, 1);
  const TS *sf = mpg_field_src(tab, src, f, (int64_t)nlev * nsrc);
  TD *df = mpg_field_dst(tab, dst, f, (int64_t)nlev * P);
  if constexpr (EPI) offset = mpg_field_off(tab, f, offset);
  const unsigned lane_bytes = (unsigned)(t & 63) * (unsigned)sizeof(TD);   // geom.h stream_store_lane
  int32_t cell[UPT];
#pragma unroll
  for (int u = 0; u < UPT; ++u) {
    int q = t + NT * u;
    cell[u] = q < nU ? cells[q] : -1;
  }
  TS pf[NPF];
#pragma unroll
  for (int lv = 0; lv < LC; ++lv)
#pragma unroll
    for (int u = 0; u < UPT; ++u) pf[lv * UPT + u] = (cell[u] >= 0 && lv < nlev) ? MPG_LDG(sf + ((int64_t)lv * nsrc + cell[u])) : (TS)0;
  for (int k0 = 0; k0 < nlev; k0 += LC) {
#pragma unroll
    for (int lv = 0; lv < LC; ++lv)
#pragma unroll
      for (int u = 0; u < UPT; ++u)
        if (cell[u] >= 0) lds[lv * nup + t + NT * u] = (double)pf[lv * UPT + u];
    for (int q = t + NT * UPT; q < nU; q += NT) {  // surplus cells of an unusually large tile
      int32_t c = cells[q];
      for (int lv = 0; lv < LC; ++lv) lds[lv * nup + q] = (k0 + lv < nlev) ? (double)sf[(int64_t)(k0 + lv) * nsrc + c] : 0.0;
    }
    __syncthreads();
    const int kn1 = k0 + LC;
    if (kn1 < nlev) {
#pragma unroll
      for (int lv = 0; lv < LC; ++lv)
#pragma unroll
        for (int u = 0; u < UPT; ++u)
          pf[lv * UPT + u] = (cell[u] >= 0 && kn1 + lv < nlev) ? MPG_LDG(sf + ((int64_t)(kn1 + lv) * nsrc + cell[u])) : (TS)0;
    }
    const int kn = min(LC, nlev - k0);
    for (int kk = 0; kk < kn; ++kk) {
      const double *row = lds + kk * nup;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        double a = row[pts.l[r][0]], b = row[pts.l[r][1]], e = row[pts.l[r][2]];
        double val = wsum3(pts.ww[r][0], a, pts.ww[r][1], b, pts.ww[r][2], e);
        val = pts.mapped[r] ? val : 0.0;
        if constexpr (EPI) val = fma(val, scale, offset);
        if (pts.act[r]) stream_store_lane((TD)val, df + (int64_t)(k0 + kk) * P + pts.off[r], lane_bytes, st_all_nt != 0);
      }
    }
    __syncthreads();
  }
}

struct CfuVariant { int rpt, nt; };   // tile = 64 x (nt * rpt / 64) points; every variant keeps 4 * nt cells in registers
static const CfuVariant g_cfu_variants[] = {
    {2, 256},   // 0: 64 x 8 points, up to 1024 cells per tile: the base shape
    {4, 256},   // 1: 64 x 16 points on 256 threads (four points each): cells shared a lot (configs 2, 5)
    {2, 512},   // 2: 64 x 16 points on 512 threads, up to 2048 cells per tile (config 4: 1591)
};
#define CFU_BASE 0
#define CFU_WIDE 1
#define CFU_TALL 2
int mpg_cfu_num_variants() { return (int)(sizeof(g_cfu_variants) / sizeof(g_cfu_variants[0])); }
static int cfu_capacity(int variant) { return 4 * g_cfu_variants[variant].nt; }
static int lfu_build_shape(mpg_handle_s *h, int txu, int tyu, hipStream_t s, int cap = 1024, int dmax = 8);
static int cfu_build(mpg_handle_s *h, int variant, hipStream_t s) {
  const CfuVariant &v = g_cfu_variants[variant];
  return lfu_build_shape(h, 64, v.nt * v.rpt / 64, s, cfu_capacity(variant), 16);
}

template <int SB>
static int launch_build(bool fill, mpg_handle_s *h, int talign, int txu, int tyu, int ntx, int64_t nblocks, int tile_step, int32_t *cnt,
                        int32_t *cells, int stride, uint16_t *lidx, unsigned long long *stats, hipStream_t s) {
  if (fill)
    k_lfu_build<true, SB><<<(unsigned)nblocks, LFU_THREADS, 0, s>>>(h->idx.p, h->nx_dst, h->ny_dst, talign, txu, tyu, ntx, tile_step, cnt, cells,
                                                                   stride, lidx, stats);
  else
    k_lfu_build<false, SB><<<(unsigned)nblocks, LFU_THREADS, 0, s>>>(h->idx.p, h->nx_dst, h->ny_dst, talign, txu, tyu, ntx, tile_step, cnt, cells,
                                                                    stride, lidx, stats);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

// The key of a set of lists: tile shape, alignment rule and the capacity they were judged against (a list that fits 2048
// cells per tile with shifted rows may have to be rebuilt unshifted for a 1024-cell kernel).
static int lists_key(int txu, int tyu, int cap, int dmax) { return (txu * 1024 + tyu) | (dmax > 8 ? 1 << 24 : 0) | (((cap >> 8) & 15) << 25); }

// tile lists for tiles of txu x tyu target points (cached in the handle, keyed by shape / alignment rule / capacity class)
static int lfu_build_shape(mpg_handle_s *h, int txu, int tyu, hipStream_t s, int cap, int dmax) {
  const int key = lists_key(txu, tyu, cap, dmax);
  if (h->ut_rpt == key) return MPG_SUCCESS;
  if (h->ut2_rpt == key) {  // the other layout's shape: swap the parked lists in, no device work
    std::swap(h->ut_cnt, h->ut2_cnt);
    std::swap(h->ut_cells, h->ut2_cells);
    std::swap(h->lidx, h->lidx2);
    std::swap(h->ut_rpt, h->ut2_rpt);
    std::swap(h->ut_max, h->ut2_max);
    std::swap(h->ut_align, h->ut2_align);
    std::swap(h->ut_total, h->ut2_total);
    std::swap(h->ut_lines, h->ut2_lines);
    std::swap(h->ut_stride, h->ut2_stride);
    std::swap(h->ut_order, h->ut2_order);
    for (int c = 0; c < 7; ++c) std::swap(h->ut_cls_off[c], h->ut2_cls_off[c]);
    return MPG_SUCCESS;
  }
  int rc;
  if (h->ut_rpt) {  // park the lists in use (dropping what was parked) and build the new shape beside them
    h->ut2_cnt.free();
    h->ut2_cells.free();
    h->lidx2.free();
    h->ut2_cnt = h->ut_cnt;
    h->ut2_cells = h->ut_cells;
    h->lidx2 = h->lidx;
    h->ut2_rpt = h->ut_rpt;
    h->ut2_max = h->ut_max;
    h->ut2_align = h->ut_align;
    h->ut2_total = h->ut_total;
    h->ut2_lines = h->ut_lines;
    h->ut2_stride = h->ut_stride;
    h->ut2_order.free();
    h->ut2_order = h->ut_order;
    h->ut_order = DevBuf<int32_t>();
    for (int c = 0; c < 7; ++c) h->ut2_cls_off[c] = h->ut_cls_off[c];
    h->ut_cnt = DevBuf<int32_t>();
    h->ut_cells = DevBuf<int32_t>();
    h->lidx = DevBuf<uint16_t>();
  }
  h->ut_cnt.free();
  h->ut_cells.free();
  h->ut_order.free();
  h->ut_rpt = 0;
  const bool big = 3 * txu * tyu > LFU_SORT;   // tiles of more than 1365 points sort in an 8192-entry buffer
  if (3 * txu * tyu > 2 * LFU_SORT) {
    mpg_set_error("staged Regrid: tile of %d x %d points exceeds the sort buffer", txu, tyu);
    return MPG_ERR_UNSUPPORTED;
  }
  // Row-shifted tiles (aligned store segments, mpg_internal.h) first; when their longest list does not fit the staged
  // kernel's cells per tile (grids whose rows start at many different offsets in a line: the shifted rows of a tile
  // then spread over up to 31 more columns) the lists are built again for unshifted tiles.
  h->ut_align = mpg_tile_align(h->nx_dst, dmax);
  if (h->nx_dst % h->ut_align == 0) h->ut_align = 1;   // every row starts aligned already: nothing to shift
  const int nty = (h->ny_dst + tyu - 1) / tyu;
  const int stride = (3 * txu * tyu + 31) & ~31;
  TmpBuf<unsigned long long> stats;
  if ((rc = stats.alloc(3, s))) return rc;
  if (!h->lidx.p && (rc = h->lidx.alloc(3 * (size_t)h->n_dst))) return rc;
  unsigned long long hs[3] = {0, 0, 0};
  for (;;) {
    const int ntx = mpg_tile_ntx(h->nx_dst, txu, h->ut_align);
    const int64_t ntile = (int64_t)ntx * nty;
    h->ut_cnt.free();
    h->ut_cells.free();
    if ((rc = h->ut_cnt.alloc(ntile + 1)) || (rc = h->ut_cells.alloc((size_t)ntile * stride + 1))) return rc;
    MPG_HIP(hipMemsetAsync(stats.p, 0, 3 * sizeof(unsigned long long), s));
    rc = big ? launch_build<2 * LFU_SORT>(true, h, h->ut_align, txu, tyu, ntx, ntile, 1, h->ut_cnt.p, h->ut_cells.p, stride, h->lidx.p, stats.p, s)
             : launch_build<LFU_SORT>(true, h, h->ut_align, txu, tyu, ntx, ntile, 1, h->ut_cnt.p, h->ut_cells.p, stride, h->lidx.p, stats.p, s);
    if (rc) return rc;
    MPG_HIP(hipMemcpyAsync(hs, stats.p, sizeof(hs), hipMemcpyDeviceToHost, s));
    MPG_HIP(hipStreamSynchronize(s));
    if (h->ut_align > 1 && (int64_t)hs[1] > cap) {
      h->ut_align = 1;
      continue;
    }
    break;
  }
  h->ut_rpt = key;
  h->ut_stride = stride;
  h->ut_max = (int)hs[1];
  h->ut_total = (int64_t)hs[2];
  h->ut_lines = (int64_t)hs[0];
  // The tiles grouped by list length (classes of at most 64, 128, ... 1024 cells), tile order kept inside a class: the staged level-fast
  // kernel is launched once per class with the row slots that class needs (round 5 chose the slots per HANDLE from its longest list: on a
  // global lat-lon grid the tiles at 60 degrees list half, at 80 degrees a fifth of the cells of a tile at the equator).
  for (int c = 0; c < 7; ++c) h->ut_cls_off[c] = 0;
  if (txu == 64 && (tyu == 8 || tyu == 16)) {   // the tile shapes of k_apply3_lfu (LFU_NT / 64 rows; 16: the -DLFU_NT=1024 experiment)
    const int ntx = mpg_tile_ntx(h->nx_dst, txu, h->ut_align);
    const int64_t ntile = (int64_t)ntx * nty;
    std::vector<int32_t> cnt((size_t)ntile), order((size_t)ntile);
    MPG_HIP(hipMemcpyAsync(cnt.data(), h->ut_cnt.p, sizeof(int32_t) * (size_t)ntile, hipMemcpyDeviceToHost, s));
    MPG_HIP(hipStreamSynchronize(s));
    auto cls_of = [](int n) {
      int c = 0;
      while ((64 << c) < n && c < 5) ++c;
      return c;   // 5: more than 1024 cells (no staged level-fast kernel holds them)
    };
    int64_t n_in[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t t = 0; t < ntile; ++t) n_in[cls_of(cnt[(size_t)t])]++;
    int64_t off[7];
    off[0] = 0;
    for (int c = 0; c < 6; ++c) off[c + 1] = off[c] + n_in[c];
    int64_t fill[6];
    for (int c = 0; c < 6; ++c) fill[c] = off[c];
    for (int64_t t = 0; t < ntile; ++t) order[(size_t)fill[cls_of(cnt[(size_t)t])]++] = (int32_t)t;
    if ((rc = h->ut_order.alloc((size_t)ntile + 1))) return rc;
    MPG_HIP(hipMemcpyAsync(h->ut_order.p, order.data(), sizeof(int32_t) * (size_t)ntile, hipMemcpyHostToDevice, s));
    MPG_HIP(hipStreamSynchronize(s));
    for (int c = 0; c < 7; ++c) h->ut_cls_off[c] = (int)off[c];
  }
  return MPG_SUCCESS;
}

int mpg_lfu_build_shape(mpg_handle_s *h, int txu, int tyu, hipStream_t s) { return lfu_build_shape(h, txu, tyu, s); }

// reuse = 3 * (points of the sampled tiles) / (their unique cells), from every `step`-th tile of the 64 x tyu tiling --
// a count-only pass over a sample costs a few per cent of a list build, and a handle that ends up with a gather kernel
// (configuration 4) never builds lists at all
static int sampled_reuse(mpg_handle_s *h, int tyu, hipStream_t s, float *reuse) {
  const int align = h->nx_dst % mpg_tile_align(h->nx_dst) == 0 ? 1 : mpg_tile_align(h->nx_dst);
  const int ntx = mpg_tile_ntx(h->nx_dst, 64, align), nty = (h->ny_dst + tyu - 1) / tyu;
  const int64_t ntile = (int64_t)ntx * nty;
  const int step = ntile > 4096 ? 13 : 1;   // odd and coprime to typical ntx: the sample walks across columns and rows
  const int64_t nb = (ntile + step - 1) / step;
  TmpBuf<unsigned long long> stats;
  int rc;
  if ((rc = stats.alloc(3, s))) return rc;
  MPG_HIP(hipMemsetAsync(stats.p, 0, 3 * sizeof(unsigned long long), s));
  if ((rc = launch_build<LFU_SORT>(false, h, align, 64, tyu, ntx, nb, step, nullptr, nullptr, 0, nullptr, stats.p, s))) return rc;
  unsigned long long hs[3];
  MPG_HIP(hipMemcpyAsync(hs, stats.p, sizeof(hs), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  // mapped points are not counted separately: unmapped ones are a rim (their tiles hold fewer cells, which only raises the
  // estimate a little on grids that stick out of the mesh)
  const double pts = (double)h->n_dst * (double)nb / (double)ntile;
  *reuse = hs[2] > 0 ? (float)(3.0 * pts / (double)hs[2]) : 0.f;
  return MPG_SUCCESS;
}

// Which level-fast kernel serves this handle?  Measured on MI355X (profiles/r01_sweep_lfu.txt), 4 fields x 55 levels:
//   target points per source cell   row gather     LDS-staged
//   1.4  (C4, 3 M cells)            4.76 TB/s      3.3 TB/s
//   2.9  (C2, 655 k cells)          2.61 TB/s      4.5 TB/s
//   2.5  (C5, global lat-lon)       2.80 TB/s      4.7 TB/s
// Staging pays when a staged row is referenced often enough; the statistic that separates the cases is
// reuse = 3 * n_dst / sum(unique cells per 64 x 4 tile): ~2.5 on C4, 5-6 on C2 / C5.
#define LFU_AUTO_MIN_REUSE 3.5f
static float g_lfu_min_reuse = LFU_AUTO_MIN_REUSE;  // "lfu_min_reuse_x10" knob (level-fast choice only; decided at a handle's first call)
void mpg_lfu_set_min_reuse_x10(int v) { g_lfu_min_reuse = 0.1f * (float)v; }
// "field_band" knob (geom.h band_map): -1 each kernel's own default, 0 field-major, > 0 tiles per band.  Measured (13 fields x
// 55 levels, profiles/r04_lf_experiments.txt): the level-fast row gather gains 4 % from bands of 1024 tiles (its 36 B of
// indices and weights per point are 7 % of its traffic and a band's 2.4 MB stay in the XCD's L2 from one field to the
// next); the staged kernels LOSE 12-17 % on configuration 5 (thirteen fields' source rows then compete for the L2 that
// serves the re-read ring of neighbouring tiles) and are level on configuration 4: they stay field-major.
static int g_staged_store = 0;   // "staged_store" knob (A/B): 0 = per lane (geom.h stream_store_lane), 2 = every lane non-temporal (rounds 2-6a)
void mpg_set_staged_store(int v) { g_staged_store = v; }
static int g_field_band = -1;
int mpg_field_band(int kernel_default) { return g_field_band < 0 ? kernel_default : g_field_band; }
void mpg_set_field_band(int v) { g_field_band = v; }

// -> *lf_variant = MPG_LF_STAGED or MPG_LF_ROWS ("lf_variant" numbering, mpg_internal.h)
int mpg_lfu_auto(mpg_handle_s *h, hipStream_t s, int *lf_variant) {
  if (h->lf_choice == 0) {
    int rc = sampled_reuse(h, 4, s, &h->lf_reuse);
    if (rc) return rc;
    h->lf_choice = h->lf_reuse >= g_lfu_min_reuse ? 1 : -1;
  }
  *lf_variant = h->lf_choice > 0 ? MPG_LF_STAGED_DEFAULT : MPG_LF_ROWS;
  return MPG_SUCCESS;
}

static void drop_lists_in_use(mpg_handle_s *h) {
  h->ut_cnt.free();
  h->ut_cells.free();
  h->lidx.free();
  h->ut_rpt = 0;
}

// cell-fast: same statistic (measured: C2 reuse 5+ -> staged 1.5x faster; C4 reuse 2.5 -> equal to the lane gather, 3-5 %
// ahead on a Morton-numbered mesh).  Per-handle choice (a3_staged = -1, the default; profiles/r01_sweep_cfu.txt,
// r02_sweep_morton.txt): 64 x 16-point lists are built once; 256 threads serve them when cells are shared a lot
// (reuse >= 3.5) and a tile holds at most 1024 cells (C2, C5), else 512 threads when a tile holds at most 2048 (C4), else
// 64 x 8-point tiles when those hold at most 1024, else the lane-gather kernel (a fine mesh under a coarse grid).
// cf_choice holds variant + 1, or -1 for the lane-gather kernel.
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
    if (h->cf_choice < 0) drop_lists_in_use(h);
  }
  *cfu_variant = h->cf_choice > 0 ? h->cf_choice - 1 : -1;
  return MPG_SUCCESS;
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
    if (h->cf_choice < 0) drop_lists_in_use(h);
  }
  *fits = h->cf_choice > 0;
  return MPG_SUCCESS;
}

template <typename TS, typename TD, int RPT, int NT, bool EPI>
static int launch_cfu(mpg_handle_s *h, const void *src, int nlev, int nfields, void *dst, double scale, double offset, hipStream_t s,
                      const FieldTab &tab) {
  constexpr int TYU = NT * RPT / 64;
  const int ntx = mpg_tile_ntx(h->nx_dst, 64, h->ut_align), nty = (h->ny_dst + TYU - 1) / TYU;
  const size_t um = h->ut_max > 0 ? h->ut_max : 1;   // unmapped points read slot 0
  size_t lds = sizeof(double) * um * 4 + 16;
  lds = std::min<size_t>(lds + (size_t)g_staged_lds_pad_kb * 1024, std::max<size_t>(lds, 160 * 1024));
  if (lds > 160 * 1024) {
    mpg_set_error("Regrid(CELL_FAST, staged): %d unique cells per tile exceed the LDS", h->ut_max);
    return MPG_ERR_UNSUPPORTED;
  }
  auto fn = k_apply3_cfu<TS, TD, RPT, NT, EPI>;
  if (lds > 48 * 1024) MPG_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  fn<<<(unsigned)ntx * nty * nfields, NT, lds, s>>>(h->ut_cnt.p, h->ut_cells.p, h->ut_stride, h->lidx.p, h->w.p, (const TS *)src, (TD *)dst, h->nx_dst,
                                                   h->ny_dst, h->ut_align, h->n_src, nlev, ntx, nty, (int)um, scale, offset, mpg_field_band(0), g_staged_store == 2, tab);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

template <int RPT, int NT>
static int launch_cfu_types(mpg_handle_s *h, const void *src, int src_f32, int nlev, int nfields, void *dst, int dst_f32, bool epi, double scale,
                            double offset, hipStream_t s, const FieldTab &tab) {
  if (!epi) return launch_cfu<double, double, RPT, NT, false>(h, src, nlev, nfields, dst, 1.0, 0.0, s, tab);
  if (src_f32 && dst_f32) return launch_cfu<float, float, RPT, NT, true>(h, src, nlev, nfields, dst, scale, offset, s, tab);
  if (src_f32) return launch_cfu<float, double, RPT, NT, true>(h, src, nlev, nfields, dst, scale, offset, s, tab);
  if (dst_f32) return launch_cfu<double, float, RPT, NT, true>(h, src, nlev, nfields, dst, scale, offset, s, tab);
  return launch_cfu<double, double, RPT, NT, true>(h, src, nlev, nfields, dst, scale, offset, s, tab);
}

// The staged cell-fast Regrid of one variant (lists built / swapped in as needed).  epi = false: mpg_regrid_dev (float64
// both sides, the result as it stands, sign of zero included).
int mpg_k_apply3_cfu(mpg_handle_s *h, int variant, const void *src, int src_f32, int nlev, int nfields, void *dst, int dst_f32, bool epi,
                     double scale, double offset, hipStream_t s, const FieldTab &tab) {
  int rc = cfu_build(h, variant, s);
  if (rc) return rc;
  if (h->ut_max > cfu_capacity(variant) && h->ut_max * 4 * sizeof(double) > 150 * 1024) return MPG_ERR_UNSUPPORTED;
  if (variant == CFU_TALL) return launch_cfu_types<2, 512>(h, src, src_f32, nlev, nfields, dst, dst_f32, epi, scale, offset, s, tab);
  if (variant == CFU_WIDE) return launch_cfu_types<4, 256>(h, src, src_f32, nlev, nfields, dst, dst_f32, epi, scale, offset, s, tab);
  return launch_cfu_types<2, 256>(h, src, src_f32, nlev, nfields, dst, dst_f32, epi, scale, offset, s, tab);
}

// ---- level-fast, level chunks --------------------------------------------------------------------------------
// float32 rows as the MPAS history file stores them ([nCells][nVertLevels], 220 bytes per cell at 55 levels) or float64
// rows, either byte order; float64 arithmetic; dst = (TD)(value * scale + offset).  64 x (NT/64)-point tiles, 16 levels
// per chunk (lanes along the levels: one 64- / 128-byte segment per row), LDS slab [row][17] in the SOURCE type (odd
// stride: conflict-free column reads; widening to float64 happens in the combine), chunk c+1 prefetched into registers
// (16 rows per thread) while chunk c is combined.
// The body is free of divergent branches on purpose: gfx950 counts loads and stores in ONE counter (vmcnt), and the
// compiler can only wait for "the prefetched rows, not the 16 stores issued after them" when it can count the memory
// instructions between the two on every path (the branchy form waited for vmcnt(0) before every chunk).  Hence: rows
// past the end of a tile's list re-load its last row (same address for the whole wave: one L1 line), levels past the end
// of a row are clamped and the last chunk starts at nlev - 16, unmapped points combine a zero row with zero weights
// (+0.0, as the masked form gave), and lanes without a target point store through an out-of-range buffer offset, which
// the hardware drops (geom.h).  All addresses are scalar base + 32-bit lane offset: no address arithmetic per access.
// Round 4 split the workgroup into producer waves (row loads -> a second slab) and consumer waves (combine + store), one
// barrier per chunk, so that no wave's in-order counter sees both kinds: bit-identical and EQUAL in time (configuration 5
// float32 6.545 against 6.517 ms, configuration 2 level; profiles/r04_lf_experiments.txt) -- the coupling of a wave's loads
// and stores is not what holds this kernel back, and the form was not kept.  Nor is occupancy: the float32 form holds 94 VGPRs
// (two workgroups = 16 waves per CU); asked to fit 80 / 64 registers (three / four workgroups, no scratch) it ran 6.39 / 6.59
// ms against 6.40-6.45.
// Round 5: NPF, the row slots a thread loads and parks per chunk (NT / 16 * NPF rows in the slab), follows the handle's longest tile list
// instead of being 16 for everyone: a tile of a coarse mesh under a fine grid lists 30-60 cells, and the branch-free form above spent as many
// instructions re-loading and re-parking the list's last row 450 times as it spent on the 512 points (a write-dominated global 0.05-degree
// target on the 655 k mesh: 7.3 ms = 0.42 of 8 TB/s, where the bare store pattern of the same tiles runs at 5.3 TB/s,
// tools/store_pattern_probe.hip).  Same arithmetic, same bits.
template <typename TS, typename TD, int NT, bool EPI, bool SWZ, int NPF>
__global__ __launch_bounds__(NT) void k_apply3_lfu(const int32_t *__restrict__ ut_cells, int stride,
                                                   const uint16_t *__restrict__ lidx, const double *__restrict__ w, const TS *__restrict__ src,
                                                   TD *__restrict__ dst, int nx, int ny, int talign, int64_t nsrc, int nlev, int ntx,
                                                   const int32_t *__restrict__ order, unsigned n_cls,
                                                   double scale, double offset, int sbe, int dbe, int band, FieldTab tab) {
  constexpr int LC = 16, LS = LC + 1, RPP = NT / LC, TY = NT / 64, ZROW = NPF * RPP;
  extern __shared__ double lds_raw[];
  TS *slab = (TS *)lds_raw;                                  // [ZROW + 1][LS]; row ZROW stays zero
  const Swz zs = make_swz(sbe), zd = make_swz(dbe);
  const int64_t P = (int64_t)nx * ny;
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  unsigned tile;
  int f;
  band_map(lin, n_cls, gridDim.x / n_cls, (unsigned)band, tile, f);   // the tiles of this launch's class, in tile order ...
  tile = (unsigned)order[tile];                                        // ... and which tile of the grid that is
  const int t = threadIdx.x, lrow = t / LC, llev = t % LC;
  const int32_t *list = ut_cells + (int64_t)tile * stride;   // padded with its last cell up to LFU_LIST_PAD entries
  // this thread's 16 rows: byte offsets of (cell, level llev) inside the field
  uint32_t rb[NPF];
  {
    const uint32_t lv = (uint32_t)min(llev, nlev - 1);
#pragma unroll
    for (int u = 0; u < NPF; ++u) rb[u] = ((uint32_t)list[lrow + u * RPP] * (uint32_t)nlev + lv) * (uint32_t)sizeof(TS);
  }
  // this thread's target point
  const int j = (int)(tile / ntx) * TY + t / 64, i = (int)(tile % ntx) * 64 + t % 64 - mpg_tile_shift(j, nx, talign);
  const bool act = i >= 0 && i < nx && j < ny;
  const int64_t p = act ? (int64_t)j * nx + i : 0;
  int l0 = lidx[p], l1 = lidx[P + p], l2 = lidx[2 * P + p];
  double w0 = w[p], w1 = w[P + p], w2 = w[2 * P + p];
  {
    const bool mapped = act && l0 != 0xFFFF;
    l0 = mapped ? l0 * LS : ZROW * LS;
    l1 = mapped ? l1 * LS : ZROW * LS;
    l2 = mapped ? l2 * LS : ZROW * LS;
    w0 = mapped ? w0 : 0.0;
    w1 = mapped ? w1 : 0.0;
    w2 = mapped ? w2 : 0.0;
  }
  const uint32_t pb = act ? (uint32_t)p * (uint32_t)sizeof(TD) : MPG_BUF_NONE;
  const BufRsrc rs = buf_rsrc(mpg_field_src(tab, src, f, (int64_t)nlev * nsrc), (uint32_t)((uint64_t)nsrc * nlev * sizeof(TS)));
  TD *dlev = mpg_field_dst(tab, dst, f, (int64_t)nlev * P);  // the level plane the next store goes to
  if constexpr (EPI) offset = mpg_field_off(tab, f, offset);
  const uint32_t plane = (uint32_t)(P * sizeof(TD));
  const int nch = (nlev + LC - 1) / LC, k0_last = max(nlev - LC, 0);
  if (t < LS) slab[ZROW * LS + t] = (TS)0;
  TS pf[NPF];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < NPF; ++u) buf_load(pf[u], rs, rb[u], (uint32_t)k0 * (uint32_t)sizeof(TS));
  };
  auto park = [&]() {
#pragma unroll
    for (int u = 0; u < NPF; ++u) slab[(lrow + u * RPP) * LS + llev] = swz<SWZ>(pf[u], zs);
  };
  auto level = [&](int col) {
    const double a = (double)slab[l0 + col], b = (double)slab[l1 + col], e = (double)slab[l2 + col];
    double val = wsum3(w0, a, w1, b, w2, e);
    if constexpr (EPI) val = fma(val, scale, offset);
    buf_store_nt(swz<SWZ>((TD)val, zd), buf_rsrc(dlev, plane), pb);   // every lane non-temporal: the per-lane form of geom.h cost this 16-times unrolled body 16 % on configuration 5's (aligned) planes, profiles/r06_plane_alignment.md
    dlev += P;
  };
  fetch(nch > 1 ? 0 : k0_last);
  park();
  __syncthreads();
  if (nch > 1) fetch(nch > 2 ? LC : k0_last);
  for (int c = 0; c + 1 < nch; ++c) {   // full chunks
#pragma unroll
    for (int kk = 0; kk < LC; ++kk) level(kk);
    __syncthreads();
    park();
    __syncthreads();
    if (c + 2 < nch) fetch(c + 3 < nch ? (c + 2) * LC : k0_last);
  }
  // last chunk: levels (nch-1)*16 .. nlev-1 sit in columns shift .. of the slab
  const int shift = (nch - 1) * LC - (nch > 1 ? k0_last : 0), kn = nlev - (nch - 1) * LC;
  l0 += shift;
  l1 += shift;
  l2 += shift;
#pragma unroll
  for (int kk = 0; kk < LC; ++kk) {
    if (kk >= kn) break;
    level(kk);
  }
}

static int g_lfu_npf = 0;   // "lfu_npf" knob: 0 = by the handle's longest tile list, 2 / 4 / 8 / 16 = that many row slots per thread (A/B)
void mpg_lfu_set_npf(int v) { g_lfu_npf = v; }

template <typename TS, typename TD, int NT, bool EPI, int NPF>
static int launch_lfu_n(mpg_handle_s *h, const void *src, int nlev, int nfields, void *dst, double scale, double offset, int sbe, int dbe,
                        hipStream_t s, const FieldTab &tab, const int32_t *order, unsigned n_cls) {
  if (n_cls == 0) return MPG_SUCCESS;
  const int ntx = mpg_tile_ntx(h->nx_dst, 64, h->ut_align);
  constexpr int ROWS = NT / 16 * NPF;   // rows of the slab
  size_t lds = sizeof(TS) * (ROWS + 1) * 17;
  lds = std::min<size_t>(lds + (size_t)g_staged_lds_pad_kb * 1024, std::max<size_t>(lds, 160 * 1024));
  static_assert(ROWS <= LFU_LIST_PAD, "the kernel reads ROWS list entries of every tile");
  auto fn = (sbe || dbe) ? k_apply3_lfu<TS, TD, NT, EPI, true, NPF> : k_apply3_lfu<TS, TD, NT, EPI, false, NPF>;
  if (lds > 48 * 1024) MPG_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  fn<<<n_cls * (unsigned)nfields, NT, lds, s>>>(h->ut_cells.p, h->ut_stride, h->lidx.p, h->w.p, (const TS *)src, (TD *)dst,
                                               h->nx_dst, h->ny_dst, h->ut_align, h->n_src, nlev, ntx, order, n_cls, scale, offset, sbe, dbe, mpg_field_band(0), tab);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}
// One launch per class of tiles (lists of at most 64, 128, 256, 512, 1024 cells: NPF = 2, 4, 8, 16, 32 row slots per thread on 512
// threads).  "lfu_npf" (A/B): a forced value serves every class it can hold, as rounds 1-5 did with 16.
template <typename TS, typename TD, int NT, bool EPI>
static int launch_lfu(mpg_handle_s *h, const void *src, int nlev, int nfields, void *dst, double scale, double offset, int sbe, int dbe,
                      hipStream_t s, const FieldTab &tab) {
  constexpr int RPP = NT / 16, NPF0 = 64 / RPP;   // class c holds 64 << c rows = RPP * (NPF0 << c)
  static_assert(RPP * NPF0 == 64, "class c holds 64 << c rows");
  if (h->ut_cls_off[6] > h->ut_cls_off[5] || !h->ut_order.p || h->ut_stride < 64 || (uint64_t)h->n_src * (uint64_t)nlev * sizeof(TS) >= 0xFFFFFFFFull ||
      (uint64_t)h->n_dst * sizeof(TD) >= 0xFFFFFFFFull)
    return MPG_ERR_UNSUPPORTED;   // a tile lists more than 1024 cells, or the 32-bit offsets do not reach
  for (int c = 0; c < 5; ++c) {
    const unsigned n_cls = (unsigned)(h->ut_cls_off[c + 1] - h->ut_cls_off[c]);
    if (n_cls == 0) continue;
    if (h->ut_stride < (64 << c)) return MPG_ERR_UNSUPPORTED;
    const int32_t *order = h->ut_order.p + h->ut_cls_off[c];
    int cc = c;                                                       // the class whose row slots serve this one
    while (cc < 4 && (NPF0 << cc) < g_lfu_npf && h->ut_stride >= (64 << (cc + 1))) ++cc;   // "lfu_npf": at least that many (A/B)
    int rc;
    switch (cc) {
      case 0: rc = launch_lfu_n<TS, TD, NT, EPI, NPF0>(h, src, nlev, nfields, dst, scale, offset, sbe, dbe, s, tab, order, n_cls); break;
      case 1: rc = launch_lfu_n<TS, TD, NT, EPI, NPF0 * 2>(h, src, nlev, nfields, dst, scale, offset, sbe, dbe, s, tab, order, n_cls); break;
      case 2: rc = launch_lfu_n<TS, TD, NT, EPI, NPF0 * 4>(h, src, nlev, nfields, dst, scale, offset, sbe, dbe, s, tab, order, n_cls); break;
      case 3: rc = launch_lfu_n<TS, TD, NT, EPI, NPF0 * 8>(h, src, nlev, nfields, dst, scale, offset, sbe, dbe, s, tab, order, n_cls); break;
      default: rc = launch_lfu_n<TS, TD, NT, EPI, NPF0 * 16>(h, src, nlev, nfields, dst, scale, offset, sbe, dbe, s, tab, order, n_cls); break;
    }
    if (rc) return rc;
  }
  return MPG_SUCCESS;
}

#ifndef LFU_NT
#define LFU_NT 512
#endif
// LFU_NT 512: 64 x 8-point tiles; float32 rows: 35 KB of LDS, four workgroups of eight waves per CU; float64: 70 KB, two

// -> MPG_ERR_UNSUPPORTED when a tile's list does not fit the slab (the caller takes the row gather)
int mpg_k_apply3_lfu_typed(mpg_handle_s *h, const void *src, int src_type, int nlev, int nfields, void *dst, int dst_type, double scale,
                           double offset, hipStream_t s, const FieldTab &tab) {
  const int sbe = (src_type & MPG_TYPE_BE) != 0, dbe = (dst_type & MPG_TYPE_BE) != 0, sf32 = src_type & MPG_TYPE_F32, df32 = dst_type & MPG_TYPE_F32;
  int rc = lfu_build_shape(h, 64, LFU_NT / 64, s, LFU_LIST_PAD);
  if (rc) return rc;
  if (sf32 && df32) return launch_lfu<float, float, LFU_NT, true>(h, src, nlev, nfields, dst, scale, offset, sbe, dbe, s, tab);
  if (sf32) return launch_lfu<float, double, LFU_NT, true>(h, src, nlev, nfields, dst, scale, offset, sbe, dbe, s, tab);
  if (df32) return launch_lfu<double, float, LFU_NT, true>(h, src, nlev, nfields, dst, scale, offset, sbe, dbe, s, tab);
  return launch_lfu<double, double, LFU_NT, true>(h, src, nlev, nfields, dst, scale, offset, sbe, dbe, s, tab);
}
int mpg_k_apply3_lfu(mpg_handle_s *h, const double *src, int nlev, int nfields, double *dst, hipStream_t s) {
  int rc = lfu_build_shape(h, 64, LFU_NT / 64, s, LFU_LIST_PAD);
  if (rc) return rc;
  return launch_lfu<double, double, LFU_NT, false>(h, src, nlev, nfields, dst, 1.0, 0.0, 0, 0, s, FieldTab());
}

// mpg_init loads this translation unit's code object ahead of its first launch (mpg_api.hip: warm_modules)
const void *mpg_anchor_k_apply_lfu() { return (const void *)&k_lfu_build<true, LFU_SORT>; }
