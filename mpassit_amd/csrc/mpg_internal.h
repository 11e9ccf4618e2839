// Internal declarations shared by the HIP translation units of libmpassit_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/mpassit_amd.h"

// Tiles of the Regrid kernels are txu x tyu target points whose x origin is shifted per grid row so that every segment of a
// row a workgroup stores starts at a multiple of A elements of the flattened [ny][nx] plane (A = 32: 128 bytes for
// float32, 256 for float64): with nx = 1800 the unshifted 64-point segments start 32 / 64 / 96 bytes into a line in three
// rows of four, and aligned segments were measured 4 % (float64 cell-fast), 5 % (float64 level-fast) and 17 % (float32
// cell-fast) faster on configuration 4 (profiles/r02_alignment.txt).  Tile column tx of row j holds
// i = tx * txu - mpg_tile_shift(j, nx) + 0 .. txu-1; points with i < 0 or i >= nx do not exist.
// The shift widens a 2-D tile: two neighbouring grid rows share their source cells, and their segments now start
// d = (nx mod A, folded to +-A/2) points apart, so the tile references about |d| / 64 more cells (C5, nx = 3600, A = 32:
// d = 16, the staged level-fast kernel lost 15 %).  mpg_tile_align picks the largest A in {32, 16, 8} whose d stays
// within 8 points: nx = 1800 -> 32 (d = 8), 3600 -> 16 (d = 0: no shift at all), 1801 -> 16 (d = 7), 150 -> 16 (d = 6).
// The tile lists of the CELL-fast staged kernels allow d <= 16 (C5 again: float32 cell-fast is 8 % faster on 128-byte
// aligned segments, float64 equal, while the level-fast staged kernels lose 5-14 % to the wider footprint).
#ifndef MPG_TILE_ALIGN
#define MPG_TILE_ALIGN 32   // the largest alignment tried (elements); -DMPG_TILE_ALIGN=1 builds without the shift (A/B runs)
#endif
static inline __host__ __device__ int mpg_tile_align(int nx, int dmax = 8) {
  for (int a = MPG_TILE_ALIGN; a >= 8; a >>= 1) {
    int d = nx % a;
    d = d < a - d ? d : a - d;
    if (d <= dmax) return a;
  }
  return MPG_TILE_ALIGN >= 8 ? 8 : 1;
}
// align < 0: the grid's own choice (mpg_tile_align); 1: no shift
static inline __host__ __device__ int mpg_tile_shift(int j, int nx, int align = -1) {
  if (align < 0) align = mpg_tile_align(nx);
  return align > 1 ? (int)(((long long)j * nx) % align) : 0;
}
static inline int mpg_tile_ntx(int nx, int txu, int align = -1) {
  if (align < 0) align = mpg_tile_align(nx);
  return (nx + (align > 1 && nx % align ? align - 1 : 0) + txu - 1) / txu;
}

#define MPG_WAVE 64

void mpg_set_error(const char *fmt, ...);
bool mpg_is_initialized();
hipStream_t mpg_setup_stream();

#define MPG_HIP(call)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      mpg_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return MPG_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

#define MPG_CHECK_INIT()                                                               \
  do {                                                                                 \
    if (!mpg_is_initialized()) {                                                       \
      mpg_set_error("mpg_init has not been called (or no HIP device): no CPU fallback"); \
      return MPG_ERR_NOT_INITIALIZED;                                                  \
    }                                                                                  \
  } while (0)

#define MPG_ARG(cond, msg)         \
  do {                             \
    if (!(cond)) {                 \
      mpg_set_error("%s", msg);    \
      return MPG_ERR_INVALID_ARG;  \
    }                              \
  } while (0)

void mpg_release_caches();   // mpg_api.hip: parked handles + scratch blocks back to the driver (an allocation failed)

// RAII-less device buffer helper (explicit free keeps object lifetimes obvious)
template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  int alloc(size_t count) {
    n = count;
    if (count == 0) return MPG_SUCCESS;
    if (hipMalloc((void **)&p, count * sizeof(T)) != hipSuccess) {
      // up to 8 parked handles and 4 GB of scratch blocks are kept only as caches: give them back and try once more
      (void)hipGetLastError();
      mpg_release_caches();
      MPG_HIP(hipMalloc((void **)&p, count * sizeof(T)));
    }
    return MPG_SUCCESS;
  }
  void free() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

// Scoped temporary: freed when it goes out of scope, so early error returns (MPG_HIP) cannot leak device memory.
// Long-lived members of meshes / grids / handles stay plain DevBuf and are freed by their owner.
// alloc(count, stream) takes the block from a per-process cache (mpg_api.hip) and gives it back without a hipFree: on
// this runtime a hipFree costs 60-220 us and synchronises, and a conservative RegridStore makes twenty of them (2.7 of its
// 8.6 ms on configuration 4).  A cached block is only handed to a request on the SAME stream it was last used on, so
// stream order alone keeps two users apart; alloc(count) without a stream is the plain hipMalloc / hipFree pair.
void *mpg_pool_get(size_t bytes, hipStream_t s);
void mpg_pool_put(void *p, size_t bytes, hipStream_t s);
void mpg_pool_release();   // mpg_finalize: everything cached goes back to the driver
template <typename T>
struct TmpBuf : DevBuf<T> {
  bool pooled = false;
  hipStream_t pool_stream = nullptr;
  size_t pool_bytes = 0;
  TmpBuf() = default;
  TmpBuf(const TmpBuf &) = delete;
  TmpBuf &operator=(const TmpBuf &) = delete;
  ~TmpBuf() { free(); }
  using DevBuf<T>::alloc;
  int alloc(size_t count, hipStream_t s) {
    free();
    this->n = count;
    if (count == 0) return MPG_SUCCESS;
    pool_bytes = count * sizeof(T);
    this->p = (T *)mpg_pool_get(pool_bytes, s);
    if (!this->p) return MPG_ERR_HIP;
    pooled = true;
    pool_stream = s;
    return MPG_SUCCESS;
  }
  void free() {
    if (pooled && this->p) {
      mpg_pool_put(this->p, pool_bytes, pool_stream);
      this->p = nullptr;
      this->n = 0;
    } else {
      DevBuf<T>::free();
    }
    pooled = false;
  }
};

// Point set on the unit sphere, SoA
struct PointSet {
  int64_t n = 0;
  DevBuf<double> x, y, z;
  int alloc(int64_t count) {
    n = count;
    int rc;
    if ((rc = x.alloc(count))) return rc;
    if ((rc = y.alloc(count))) return rc;
    return z.alloc(count);
  }
  void free() { x.free(); y.free(); z.free(); n = 0; }
};

// AABB pyramid over a structured (nx x ny) point set: level 0 = blocks of B0 x B0 points, each upper
// level merges 2x2 nodes.  box[level] is [nodes][6] = lo.xyz, hi.xyz.
#define MPG_PYR_B0 4
#define MPG_PYR_MAXLEV 16
struct Pyramid {
  int nlev = 0;
  int nx[MPG_PYR_MAXLEV], ny[MPG_PYR_MAXLEV];
  int64_t off[MPG_PYR_MAXLEV + 1];  // node offset of each level inside `box` (in nodes)
  DevBuf<double> box;
  bool built = false;
  void free() { box.free(); built = false; nlev = 0; }
};
struct PyramidView {  // passed by value to kernels
  int nlev;
  int nx[MPG_PYR_MAXLEV], ny[MPG_PYR_MAXLEV];
  int64_t off[MPG_PYR_MAXLEV + 1];
  const double *box;
};

// Morton-sorted site BVH (leaf = 8 consecutive sorted sites, fan-out 8)
#define MPG_BVH_LEAF 8
#define MPG_BVH_FAN 8
#define MPG_BVH_MAXLEV 12
struct SiteBvh {
  int64_t n = 0;
  PointSet sorted;          // sites in Morton order
  DevBuf<int32_t> sorted_id;  // original cell id of each sorted site
  int nlev = 0;
  int64_t nnodes[MPG_BVH_MAXLEV];
  int64_t off[MPG_BVH_MAXLEV + 1];
  DevBuf<double> box;  // [nodes][6]
  bool built = false;
  void free() { sorted.free(); sorted_id.free(); box.free(); built = false; }
};
struct SiteBvhView {
  int64_t n;
  const double *sx, *sy, *sz;
  const int32_t *sid;
  int nlev;
  int64_t nnodes[MPG_BVH_MAXLEV];
  int64_t off[MPG_BVH_MAXLEV + 1];
  const double *box;
};

struct mpg_handle_s;
struct mpg_grid_s;
typedef std::tuple<void *, int, void *, int, int> HandleKey;

struct mpg_mesh_s {
  int64_t nCells = 0, nVertices = 0;
  int maxEdges = 0;
  PointSet cell, vert;          // unit vectors of cell centres / vertices
  DevBuf<int32_t> voc;          // [nCells][maxEdges] 1-based, 0-padded (as given)
  DevBuf<int32_t> tri;          // [3][nVertices] dual triangles (cells), -1 = none; CCW
  int64_t nTriValid = 0;
  DevBuf<int32_t> fan;          // [3][nCells*(maxEdges-2)] fan triangles of the Voronoi polygons (vertex ids), lazily
  int fan_origin = 0;           // the "node_fan_origin" value `fan` was built for
  SiteBvh bvh;
  // source window per mesh location (ELEMENT, NODE): Regrid sources hold ids [win_first, win_first + win_count) only and
  // every handle of this mesh indexes relative to win_first (mpg_mesh_set_source_window); whole mesh by default
  int64_t win_first[2] = {0, 0}, win_count[2] = {-1, -1};
  // GEOMETRY window (mpg_mesh_create_window; the whole mesh after mpg_mesh_create): `voc` holds the rows of cells
  // [cw0, cw0 + cwn) only, `vert` and `tri` the vertices [vw0, vw0 + vwn) only; `cell` always holds every cell centre
  // (handles, triangles and the BVH carry GLOBAL cell ids).  Kernels index the windowed arrays through the biased
  // pointers below with global ids, or loop over the local ranges.
  int64_t cw0 = 0, cwn = 0, vw0 = 0, vwn = 0;
  mpg_grid_s *geo_grid = nullptr;   // the grid the window was cut for (Stores onto any other grid are refused); nullptr: whole mesh
  bool geo_grid_gone = false;       // that grid has been destroyed
  double geo_margin = 0.0;          // chord distance from the grid within which every cell is present
  int max_valence = -1;             // most vertices any resident cell has (<= maxEdges, the width of verticesOnCell); -1 = not counted yet
  bool bvh_whole = true;            // the BVH covers every cell (false: the cells of the window only)
  const int32_t *voc_g() const { return voc.p - cw0 * maxEdges; }
  const double *vx_g() const { return vert.x.p - vw0; }
  const double *vy_g() const { return vert.y.p - vw0; }
  const double *vz_g() const { return vert.z.p - vw0; }
};

// projection constants handed to the kernels by value (proj_info subset, module_map_utils.F90:140-192)
struct ProjDev {
  int code;
  double hemi, truelat1, truelat2, stdlon, cone, polei, polej, rebydx, lat1, lon1, knowni, knownj, latinc, loninc;
  double rsw, dlon;   // PROJ_PS / PROJ_MERC (set_ps, set_merc)
  int nxmin, nxmax;
};

struct mpg_grid_s {
  int nx = 0, ny = 0, periodic = 0;
  PointSet pts[4];  // indexed by MPG_STAGGERLOC_*
  int snx[4], sny[4];
  Pyramid pyr[4];   // point pyramids (per stagger), built lazily
  Pyramid cellpyr;  // pyramid over CENTER cells bounded by CORNER points (conservative)
  // grids created from a projection (mpg_grid_create_proj) also keep what the output file needs
  bool from_proj = false;
  int proj_code = 0;
  ProjDev proj;                   // the projection itself: Stores on such a grid find a triangle's / a cell's target points through its
                                  // inverse in O(1) (k_target_grid.hip mpg_k_points_ij) instead of descending the box pyramid
  bool has_inverse = false;       // `proj` is set and checked against the grid's own points (mpg_grid_create_proj, mpg_grid_attach_proj)
  bool inverse_ok[4] = {true, true, true, true};   // ... per stagger: a grid made from ARRAYS may carry staggers that are not the projection's
                                  // (a file-defined grid's CORNER points come from get_cell_corners, model_grid.F90:1902-1972, a cell off)
  int proj_row0 = 0;              // the grid's first row is row proj_row0 of the projection's grid (a rank's row block)
  DevBuf<double> lon[4], lat[4];  // degrees, per stagger
  DevBuf<double> mapfac[3];       // CENTER, EDGE1, EDGE2
  DevBuf<double> cosa, sina;      // CENTER, PROJ_LC only
};

enum { MPG_KIND_FIXED = 0, MPG_KIND_CSR = 1 };
struct mpg_handle_s {
  int kind = MPG_KIND_FIXED;
  int method = 0;
  int nnz_per_row = 0;  // 3, 4, 1 or 0 (CSR)
  int64_t n_src = 0, n_dst = 0;
  int nx_dst = 0, ny_dst = 0;
  DevBuf<int32_t> idx;  // [nnz_per_row][n_dst] SoA, -1 = unmapped
  DevBuf<double> w;     // [nnz_per_row][n_dst] SoA (absent for nearest)
  DevBuf<int32_t> rowptr;  // CSR [n_dst+1]
  DevBuf<int32_t> col;
  DevBuf<double> val;
  int64_t nnz = 0;
  int refcount = 1;
  bool cached = false;
  uint64_t parked_at = 0;   // release order of a handle waiting in the cache with refcount 0 (mpg_api.hip)
  HandleKey key;
  float store_ms = 0.f;
  int64_t store_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // which branches the Store took (mpg_handle_store_stats; [0] is store_path)
  int store_path = 0;       // candidate search of the Store: 0 hierarchical (pyramid walk / BVH), 1 the grid's index space, 2 index space + BVH for the rest
  bool localized = false;
  // [first, end) of the source indices the handle references, in its CURRENT index space, once somebody has asked (the host-array
  // Regrids ask on every call: mpg_hostpipe.hip); dropped whenever the indices are rewritten (rebase, localize, source windows)
  bool src_range_valid = false;
  int64_t src_range_first = 0, src_range_end = 0;
  // pole caps of a periodic (monopole) source grid: destination point pole_dst[q] adds pole_w[q] * mean of the
  // pole_len sources starting at pole_src0[q].  Dense over the candidate rows, pole_w == 0 where not in a cap.
  int64_t n_pole = 0;
  int pole_len = 0;
  DevBuf<int32_t> pole_dst, pole_src0;
  DevBuf<double> pole_w;
  // per-tile unique source cells of the LDS-staged level-fast Regrid (k_apply_lfu.hip), built on first use
  int ut_rpt = 0, ut_max = 0;
  int ut_align = MPG_TILE_ALIGN;   // row shift of the tiles these lists were built for (1: none -- the shifted tiles' lists did not fit)
  int64_t ut_total = 0;
  int64_t ut_lines = 0, ut2_lines = 0;   // sum over the tiles of the distinct groups of 16 consecutive cell ids in their lists
  int lf_choice = 0;      // level-fast kernel picked for this handle: 0 undecided, 1 LDS-staged, -1 row-gather (from a sampled statistic)
  int cf_choice = 0;      // cell-fast kernel picked for this handle: 0 undecided, 1 LDS-staged, -1 lane-gather
  int cf_for = -99;       // "a3_staged" knob value the choice was made for
  float lf_reuse = 0.f;   // 3 * n_dst / (sum of the tiles' unique cells): references per staged row
  DevBuf<int32_t> ut_cnt, ut_cells;   // list of tile t: ut_cells[t * ut_stride .. + ut_cnt[t])
  int ut_stride = 0, ut2_stride = 0;
  DevBuf<uint16_t> lidx;  // [3][n_dst] positions in the tile's list, 0xFFFF = unmapped
  // the tiles of the staged level-fast kernel grouped by the length of their lists (k_apply_lfu.hip, round 6): ut_order = tile numbers,
  // class c (lists of at most 64 << c cells) at [ut_cls_off[c], ut_cls_off[c + 1]); one launch per class, each with the row slots it needs
  DevBuf<int32_t> ut_order, ut2_order;
  int ut_cls_off[7] = {0, 0, 0, 0, 0, 0, 0}, ut2_cls_off[7] = {0, 0, 0, 0, 0, 0, 0};
  // a second, parked set of tile lists: a job that alternates layouts on one handle (2-D fields cell-fast, 3-D fields
  // in file order) needs two tile shapes in turn; the lists of the shape not in use wait here and are swapped back in
  // instead of being rebuilt (a rebuild allocates and synchronises, which would also break hipGraph capture)
  int ut2_rpt = 0, ut2_max = 0, ut2_align = MPG_TILE_ALIGN;
  int64_t ut2_total = 0;
  DevBuf<int32_t> ut2_cnt, ut2_cells;
  DevBuf<uint16_t> lidx2;
  void free_tile_lists() {
    ut_cnt.free(); ut_cells.free(); lidx.free(); ut2_cnt.free(); ut2_cells.free(); lidx2.free(); ut_order.free(); ut2_order.free();
    ut_rpt = ut2_rpt = 0;
  }
};

void mpg_lfu_set_min_reuse_x10(int v);  // k_apply_lfu.hip
int mpg_field_band(int kernel_default);   // "field_band" knob, or the calling kernel's default when the knob is -1
void mpg_set_field_band(int v);
void mpg_lfu_set_npf(int v);                // "lfu_npf" knob (k_apply_lfu.hip)
void mpg_set_staged_store(int v);            // "staged_store" knob (k_apply_lfu.hip; A/B only)
void mpg_set_lf_rows_store(int v);          // "lf_rows_store" knob (k_apply_typed.hip; A/B only)
void mpg_set_staged_lds_pad_kb(int v);      // "staged_lds_pad_kb" knob (k_apply_lfu.hip; A/B only)
int mpg_staged_lds_pad_kb();
bool mpg_handle_is_windowed(const mpg_handle_s *h);   // mpg_api.hip: its mesh carries a source window (indices are window-relative)
void mpg_cache_detach(mpg_handle_s *h);  // mpg_api.hip: a handle about to be re-indexed in place leaves the Store cache
void mpg_hostpipe_release();  // mpg_hostpipe.hip: device slots / streams of the host-pointer Regrid pipeline, dropped by mpg_finalize
void mpg_fileio_release();  // mpg_fileio.hip: staging buffers / streams of mpg_file_to_dev, dropped by mpg_finalize

// ---- launchers implemented in the kernel TUs ---------------------------------------------------
int mpg_k_mesh_coords(int64_t n, const double *lon_rad, const double *lat_rad, PointSet &out, hipStream_t s);
int mpg_k_grid_coords(int64_t n, const double *lon_deg, const double *lat_deg, PointSet &out, hipStream_t s);
int mpg_k_target_grid(const mpg_proj *proj, mpg_grid_s *g, hipStream_t s);
// (i, j) of n points on the unit sphere in the 0-based CENTER index space of a projection-built grid, float2 per point; NaN where
// the inverse projection is not safely usable (near its pole / cut, other projections): callers fall back to the pyramid there.
// false: this grid has no usable inverse at all (made from arrays, or a projection without one here)
bool mpg_grid_has_inverse(const mpg_grid_s *g, int stagger = MPG_STAGGERLOC_CENTER);   // ... whose points of `stagger` sit where the projection puts them
int mpg_k_attach_proj(mpg_grid_s *g, const mpg_proj *proj, int row0, hipStream_t s);
#define MPG_LATLON_BOX_LIMIT 85.0   // degrees: up to here a lat-lon grid's index boxes serve the Stores (their pad follows the figure's latitude)
int mpg_k_points_ij(const mpg_grid_s *g, int64_t n, const double *x, const double *y, const double *z, float *ij, hipStream_t s, double latlon_limit = MPG_LATLON_BOX_LIMIT,
                    bool unwrap_i = false);
// safety margin (index units) around the index-space box of a figure whose vertices span `extent` index units
double mpg_grid_box_pad_latlon(const mpg_grid_s *g);
double mpg_grid_box_pad_coef(const mpg_grid_s *g);
double mpg_grid_box_emax(const mpg_grid_s *g);
double mpg_grid_min_index_chord(const mpg_grid_s *g, double lat_lo, double lat_hi, double margin);
int mpg_k_dual_triangles(mpg_mesh_s *m, hipStream_t s);
int mpg_k_tri_scatter(mpg_mesh_s *m, int32_t *cnt, hipStream_t s);
int mpg_k_tri_canon(mpg_mesh_s *m, const int32_t *cnt, hipStream_t s);
// k_mesh_window.hip: cuts the mesh to what grid `g` can see (fills cw0 .. geo_margin, voc, vert, tri of `m`, whose `cell` holds all centres)
int mpg_k_mesh_window(mpg_mesh_s *m, mpg_grid_s *g, const double *latVertex, const double *lonVertex, const int32_t *verticesOnCell, hipStream_t s);
int mpg_k_voc_check(const int32_t *voc_dev, int64_t nent, int64_t nV, const char *who, hipStream_t s);
int mpg_k_mesh_coords_dev(int64_t n, const double *lon_rad_dev, const double *lat_rad_dev, double *x, double *y, double *z, unsigned long long *bad_dev,
                          hipStream_t s);
int mpg_k_build_pyramid(const PointSet &pts, int nx, int ny, Pyramid &pyr, hipStream_t s);
int mpg_k_build_cell_pyramid(const PointSet &corner, int nx, int ny, Pyramid &pyr, hipStream_t s);
PyramidView mpg_pyr_view(const Pyramid &p);
int mpg_k_store_bilinear_mesh(mpg_mesh_s *m, mpg_grid_s *g, int stagger, int meshloc, mpg_handle_s *h, hipStream_t s);
int mpg_k_store_nearest(mpg_mesh_s *m, mpg_grid_s *g, int stagger, mpg_handle_s *h, hipStream_t s);
int mpg_k_store_conserve(mpg_mesh_s *m, mpg_grid_s *g, mpg_handle_s *h, hipStream_t s);
int mpg_k_store_grid_bilinear(mpg_grid_s *g, int dst_stagger, mpg_handle_s *h, hipStream_t s);
int mpg_k_build_bvh(mpg_mesh_s *m, hipStream_t s, bool whole = true);
int mpg_k_apply(mpg_handle_s *h, const double *src, int layout, int nlev, int nfields, double *dst, hipStream_t s);
// The fields of a bundle as separate allocations (mpg_regrid_bundle_typed_dev: an ESMF field bundle holds separate arrays): up
// to MPG_TAB_MAX device pointers on either side and one epilogue offset per field, handed to the Regrid kernels BY VALUE -- in
// their argument block, so there is no table in device memory to allocate, to keep alive or to race on, and a launch can be
// captured in a hipGraph like any other; a longer bundle goes out as several launches of MPG_TAB_MAX fields.  n = 0: the fields
// are consecutive slabs behind src / dst, as everywhere else.  A kernel reads its own field's entries (scalar loads, the field
// index is uniform in a workgroup) where it would otherwise have added f * slab to the base.
#define MPG_TAB_MAX 32
struct FieldTab {
  const void *src[MPG_TAB_MAX];
  void *dst[MPG_TAB_MAX];
  double off[MPG_TAB_MAX];
  int n;
  __host__ __device__ FieldTab() : src{}, dst{}, off{}, n(0) {}
};
template <typename TS>
__device__ __forceinline__ const TS *mpg_field_src(const FieldTab &t, const TS *src, int f, int64_t slab) {
  return t.n ? (const TS *)t.src[f] : src + (int64_t)f * slab;
}
template <typename TD>
__device__ __forceinline__ TD *mpg_field_dst(const FieldTab &t, TD *dst, int f, int64_t slab) {
  return t.n ? (TD *)t.dst[f] : dst + (int64_t)f * slab;
}
__device__ __forceinline__ double mpg_field_off(const FieldTab &t, int f, double offset) { return t.n ? t.off[f] : offset; }
// src_type / dst_type below: MPG_TYPE_F64 / MPG_TYPE_F32, optionally | MPG_TYPE_BE (include/mpassit_amd.h)
int mpg_k_apply_typed(mpg_handle_s *h, const void *src, int src_type, int layout, int nlev, int nfields, void *dst, int dst_type,
                      double scale, double offset, hipStream_t s, const FieldTab &tab = FieldTab());
// "lf_variant" numbering of the level-fast 3-point Regrid
enum { MPG_LF_ROWS = 0, MPG_LF_STAGED = 1, MPG_LF_ROWTILES = 2 };
#ifndef MPG_LF_STAGED_DEFAULT
#define MPG_LF_STAGED_DEFAULT MPG_LF_STAGED   // what the per-handle choice takes when staging pays
#endif
// a bundle of fewer levels than this (nlev * nfields) is served by the gather kernels in the per-handle modes: building
// tile lists costs more than every 2-D field of a job together (profiles/r02j: 4.9 ms of list builds in a cold
// configuration-4 job), and one level gives a staged kernel nothing to amortise its prologue over
#define MPG_STAGE_MIN_LEVELS 8
int mpg_cfu_num_variants();
int mpg_k_apply3_cfu(mpg_handle_s *h, int variant, const void *src, int src_f32, int nlev, int nfields, void *dst, int dst_f32, bool epi,
                     double scale, double offset, hipStream_t s, const FieldTab &tab = FieldTab());
int mpg_k_apply3_lfu(mpg_handle_s *h, const double *src, int nlev, int nfields, double *dst, hipStream_t s);
int mpg_k_apply3_lfu_typed(mpg_handle_s *h, const void *src, int src_type, int nlev, int nfields, void *dst, int dst_type, double scale,
                           double offset, hipStream_t s, const FieldTab &tab = FieldTab());
int mpg_k_apply3_lf_rows(mpg_handle_s *h, const double *src, int nlev, int nfields, double *dst, hipStream_t s);
int mpg_a3_staged();  // current "a3_staged" knob
int mpg_lf_variant(); // current "lf_variant" knob
int mpg_lfu_build_shape(mpg_handle_s *h, int txu, int tyu, hipStream_t s);  // tile lists for txu x tyu-point tiles (cached per handle)
int mpg_cfu_fits(mpg_handle_s *h, int variant, hipStream_t s, int *fits);
int mpg_cfu_auto(mpg_handle_s *h, hipStream_t s, int *cfu_variant);  // -> variant index or -1 (use k_apply3_cf)
int mpg_lfu_auto(mpg_handle_s *h, hipStream_t s, int *lf_variant);   // -> MPG_LF_ROWS or the staged default
int mpg_k_pole_fix(mpg_handle_s *h, const void *src, int src_type, int layout, int nlev, int nfields, void *dst, int dst_type,
                   double scale, double offset, hipStream_t s, const FieldTab &tab = FieldTab());
int mpg_k_bswap(void *buf, int64_t n, int elem_size, hipStream_t s);
int mpg_k_post_cast(const double *src, int64_t n, double scale, double offset, float *dst, int dst_be, hipStream_t s);
int mpg_k_post_layer_mean(const double *src, int nlevp1, int64_t P, float *dst, int dst_be, hipStream_t s);
int mpg_k_post_ptop(const double *src, int nlev, int64_t P, double *ptop_host, hipStream_t s);
int mpg_k_post_ptop_parts(const double *src, int nlev, int64_t P, double *vmax_host, double *candmin_host, int *has_cand_host, hipStream_t s);
int mpg_k_rotate(int64_t npts, int nlev, const double *cosa, const double *sina, double *u, double *v, hipStream_t s);
int mpg_k_wind_destagger(mpg_handle_s *h1, mpg_handle_s *h2, const double *cosa, const double *sina, const double *um, const double *vm, int nlev,
                         void *u, void *v, int dst_type, double *um_rot, double *vm_rot, hipStream_t s);   // k_wind.hip
int mpg_k_pack(const double *src, int64_t n_src, int nlev, const int32_t *ids, int64_t n_ids, double *dst, hipStream_t s);
int mpg_k_tune(const char *key, int value);
int mpg_store_boxes();         // "store_boxes" knob: 1 (default) index-space candidate boxes on projection-built grids, 0 pyramid walk only
int mpg_bilinear_linetype();   // "bilinear_linetype" knob: 0 ray from the centre (default), 1 along the triangle's normal
int mpg_node_fan_origin();     // "node_fan_origin" knob: apex of a polygon's fan = listed vertex number (value mod n); 0 (default) the first, -1 the last
int mpg_grid_inside_tol_exp(); // "grid_inside_tol_exp" knob: Grid -> Grid inside tolerance 10^-value (default 10 = MPG_TOL)
int mpg_nearest_variant();
void mpg_set_nearest_variant(int v);
int mpg_k_rebase(mpg_handle_s *h, int64_t base, int64_t n_local, hipStream_t s, bool keep_global = false);
int mpg_k_source_range(mpg_handle_s *h, int64_t *first, int64_t *end, hipStream_t s);
int mpg_k_unique_sources(mpg_handle_s *h, std::vector<int32_t> &ids, bool remap, hipStream_t s);
// k_prims.hip / k_sort.hip: the device-wide primitives of the Stores
int mpg_scan_excl_i32(const int32_t *in, int32_t *out, int64_t n, hipStream_t s);     // out[i] = in[0] + .. + in[i - 1]
int mpg_sum_i32_i64(const int32_t *in, int64_t n, long long *out_dev, hipStream_t s);
int mpg_sort_pairs_u64_i32(const unsigned long long *keys_in, unsigned long long *keys_out, const int32_t *vals_in, int32_t *vals_out, int64_t n,
                           hipStream_t s);
