// C-ABI entry points of libmpassit_amd.so (declared in include/mpassit_amd.h).
// Thin glue: argument checks, object lifetime, handle cache, host<->device staging.  All arithmetic is
// in the k_*.hip kernels; there is deliberately no CPU code path.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "mpg_internal.h"

static thread_local char g_err[1024] = "";
static bool g_init = false;
static int g_device = -1;
static hipStream_t g_stream = nullptr;
static std::map<HandleKey, mpg_handle_s *> g_cache;
// Released handles stay in the cache ("parked", refcount 0) until their mesh or grid is destroyed, the library is finalized
// or more than MPG_MAX_PARKED of them wait: the reference stores the SAME element -> CENTER bilinear weights in
// interp_diag_data and again in interp_hist_data with a release in between (interp.F90:123,148 and :207), and a second
// mpg_regrid_store of a released 5-tuple was a second run of the rasteriser (r02j: 2 x k_tri_raster in one job).
#define MPG_MAX_PARKED 8
static uint64_t g_park_clock = 0;
// The cache is shared with the Store worker below (it inserts finished handles): every walk, lookup and change holds this lock.
static std::recursive_mutex g_cache_mu;
#define CACHE_LOCK() std::lock_guard<std::recursive_mutex> cache_lock_(g_cache_mu)
static void store_worker_drain();   // below: every queued / running Store has finished when this returns
static void store_worker_stop();
static int store_worker_start(int device);

void mpg_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
bool mpg_is_initialized() { return g_init; }
hipStream_t mpg_setup_stream() { return g_stream; }
int mpg_device_index() { return g_device; }

void mpg_cache_detach(mpg_handle_s *h) {
  CACHE_LOCK();
  if (h->cached) {
    g_cache.erase(h->key);
    h->cached = false;
  }
}

// ---- cache of temporary device blocks (TmpBuf::alloc(count, stream), mpg_internal.h) ------------------------------------
namespace {
struct PoolBlock { void *p; size_t bytes; hipStream_t stream; };
std::vector<PoolBlock> g_pool;
std::mutex g_pool_mu;
size_t g_pool_bytes = 0;
const size_t POOL_CAP_BYTES = (size_t)4 << 30;   // beyond this a returned block goes straight back to the driver
}  // namespace

void *mpg_pool_get(size_t bytes, hipStream_t s) {
  bytes = (bytes + 255) & ~(size_t)255;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int best = -1;
    for (int i = 0; i < (int)g_pool.size(); ++i) {
      const PoolBlock &b = g_pool[i];
      if (b.stream != s || b.bytes < bytes || b.bytes > 2 * bytes + (1 << 20)) continue;   // same stream only; bounded waste
      if (best < 0 || b.bytes < g_pool[best].bytes) best = i;
    }
    if (best >= 0) {
      void *p = g_pool[best].p;
      g_pool_bytes -= g_pool[best].bytes;
      g_pool.erase(g_pool.begin() + best);
      return p;
    }
  }
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {   // the cache may be what fills the device: drop it and try once more
    mpg_pool_release();
    e = hipMalloc(&p, bytes);
  }
  if (e != hipSuccess) {
    mpg_set_error("HIP error %s allocating %zu bytes of scratch", hipGetErrorString(e), bytes);
    return nullptr;
  }
  return p;
}

// The size a block is cached under is the size it was ASKED with, rounded as mpg_pool_get rounds: a block found in the
// cache for a smaller request keeps its real size only if the caller returns it under that request's size -- so the cache
// tracks the real size itself.
static std::unordered_map<void *, size_t> g_pool_real;

void mpg_pool_put(void *p, size_t bytes, hipStream_t s) {
  if (!p) return;
  bytes = (bytes + 255) & ~(size_t)255;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  auto it = g_pool_real.find(p);
  if (it != g_pool_real.end()) bytes = it->second;
  else g_pool_real[p] = bytes;
  if (g_pool_bytes + bytes > POOL_CAP_BYTES) {
    g_pool_real.erase(p);
    (void)hipFree(p);
    return;
  }
  g_pool.push_back(PoolBlock{p, bytes, s});
  g_pool_bytes += bytes;
}

void mpg_pool_release() {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (const PoolBlock &b : g_pool) (void)hipFree(b.p);
  g_pool.clear();
  g_pool_real.clear();
  g_pool_bytes = 0;
}


// ---- first-call costs paid at mpg_init ---------------------------------------------------------------------------------
// The HIP runtime loads a translation unit's code object at the first launch of one of its kernels: measured on MI355X
// (tools/first_call_probe.py, profiles/r04_first_call.txt) the first nearest Store of a process took 8.0 ms against 3.1 ms
// for the second, the first conservative Store 14.5 against 4.5, the first mpg_mesh_create 12.5 against 4.5 -- MPASSIT is
// a single-shot tool (mpassit.F90:105-137), so the first values are what a run pays.  mpg_init therefore starts a helper
// thread that asks the runtime for one kernel of every translation unit (hipFuncGetAttributes loads the module) and
// creates the pinned staging of the host-array upload path, while the caller goes on (reading its namelist, opening its
// files); a call that needs a module before the helper got to it simply loads it itself (the runtime serialises that).
#define MPG_ANCHORS(X) X(k_setup) X(k_target_grid) X(k_store_bilinear) X(k_store_nearest) X(k_store_conserve) X(k_store_gridbil) \
  X(k_apply) X(k_apply_lfu) X(k_apply_typed) X(k_wind) X(k_pole) X(k_post) X(k_halo) X(mpg_comm) X(k_mesh_window) X(k_prims) X(k_sort)
#define X(n) const void *mpg_anchor_##n();
MPG_ANCHORS(X)
#undef X
static std::thread *g_warm = nullptr;   // on the heap: a process that exits without mpg_finalize must not meet ~thread of a joinable thread
// MPG_INIT_TRACE=1: where mpg_init's time and the helper thread's go, one line per step on stderr (tools/first_call_probe.py;
// profiles/r05_init_breakdown.md)
static bool g_init_trace = false;
static double now_ms() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}
static void warm_modules(int device) {
  if (hipSetDevice(device) != hipSuccess) return;
  (void)store_worker_start(device);   // the Store worker's thread and stream exist before the first RegridStore asks for them
  hipFuncAttributes a;
  double t0 = now_ms();
#define X(n)                                                                                                          \
  (void)hipFuncGetAttributes(&a, mpg_anchor_##n());                                                                   \
  if (g_init_trace) {                                                                                                 \
    const double t1 = now_ms();                                                                                       \
    fprintf(stderr, "mpg_init trace: helper thread: code object of %-18s %8.2f ms\n", #n, t1 - t0);                   \
    t0 = t1;                                                                                                          \
  }
  MPG_ANCHORS(X)
#undef X
  // the runtime builds its staging for copies from pageable host memory at the first such copy (9.5 ms in front of the first
  // mpg_mesh_create, rocprofv3 --hip-trace): one throw-away 32 MB upload does that here
  const size_t n = (size_t)32 << 20;
  void *h = malloc(n), *d = nullptr;
  hipStream_t s = nullptr;
  if (h && hipMalloc(&d, n) == hipSuccess && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess) {
    memset(h, 0, n);
    (void)hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s);
    (void)hipStreamSynchronize(s);
  }
  if (s) (void)hipStreamDestroy(s);
  if (d) (void)hipFree(d);
  free(h);
  if (g_init_trace) fprintf(stderr, "mpg_init trace: helper thread: first pageable 32 MB upload + its buffers %8.2f ms\n", now_ms() - t0);
}
static void warm_join() {   // mpg_finalize, and atexit (registered after the HIP runtime's own handlers: runs before them)
  if (g_warm) {
    if (g_warm->joinable()) g_warm->join();
    delete g_warm;
    g_warm = nullptr;
  }
}

// Everything the library holds only as a cache goes back to the driver: parked handles, scratch blocks.  Called when an
// allocation fails, before it is tried once more (DevBuf::alloc, mpg_dev_alloc).
static void drop_parked(void *obj);
static bool g_cache_busy = false;   // a caller is walking g_cache (mpg_mesh_set_source_window): its entries must stay
void mpg_release_caches() {
  if (!g_cache_busy) drop_parked(nullptr);
  mpg_pool_release();
}

extern "C" {

const char *mpg_last_error(void) { return g_err; }

int mpg_init(int device) {
  int n = 0;
  {
    const char *tr = getenv("MPG_INIT_TRACE");
    g_init_trace = tr && *tr == '1';
  }
  const double t_0 = now_ms();
  hipError_t e = hipGetDeviceCount(&n);
  const double t_1 = now_ms();
  if (e != hipSuccess || n <= 0) {
    mpg_set_error("mpg_init: no HIP device available (%s); this library has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return MPG_ERR_NOT_INITIALIZED;
  }
  MPG_ARG(device >= 0 && device < n, "mpg_init: device index out of range");
  if (g_init && device != g_device) {
    // the set-up stream, the file-io lanes and their pinned buffers belong to the first device; objects created there would
    // be paired with a stream of another device.  One device per process (one process per GPU); finalize first to switch.
    mpg_set_error("mpg_init: already initialised on device %d; call mpg_finalize before initialising device %d", g_device, device);
    return MPG_ERR_INVALID_ARG;
  }
  MPG_HIP(hipSetDevice(device));
  const double t_2 = now_ms();
  if (!g_init) {
    MPG_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    if (g_init_trace)
      fprintf(stderr, "mpg_init trace: hipGetDeviceCount (runtime start) %.2f ms, hipSetDevice %.2f ms, stream %.2f ms\n", t_1 - t_0, t_2 - t_1,
              now_ms() - t_2);
    const char *e = getenv("MPG_NO_WARMUP");   // A/B runs of the first-call probe
    if (!(e && *e == '1')) {
      g_warm = new std::thread(warm_modules, device);
      static bool registered = false;
      if (!registered) {
        atexit(warm_join);
        registered = true;
      }
    }
  }
  g_device = device;
  g_init = true;
  return MPG_SUCCESS;
}

// The helper thread of mpg_init allocates, frees and copies while it runs; a caller that starts a GLOBAL-mode stream capture right
// after mpg_init (hipStreamBeginCapture forbids such calls from other threads while it lasts) waits for it here first.
int mpg_warmup_wait(void) {
  warm_join();
  store_worker_drain();   // ... and for the Store worker (mpg_regrid_store_begin): it allocates too
  return MPG_SUCCESS;
}

static void handle_free(mpg_handle_s *h);
static void drop_parked(void *obj) {   // obj == nullptr: all of them
  CACHE_LOCK();
  for (auto it = g_cache.begin(); it != g_cache.end();) {
    mpg_handle_s *h = it->second;
    if (h->refcount == 0 && (!obj || std::get<0>(it->first) == obj || std::get<2>(it->first) == obj)) {
      it = g_cache.erase(it);
      handle_free(h);
    } else {
      ++it;
    }
  }
}

int mpg_finalize(void) {
  if (!g_init) return MPG_SUCCESS;
  warm_join();
  store_worker_stop();
  drop_parked(nullptr);
  mpg_fileio_release();
  mpg_hostpipe_release();
  (void)hipStreamSynchronize(g_stream);
  mpg_pool_release();
  (void)hipStreamDestroy(g_stream);
  g_stream = nullptr;
  g_init = false;
  return MPG_SUCCESS;
}

// GPUs this process can see; may be called before mpg_init (a launcher's ranks pick their device with it)
int mpg_device_count(int *n) {
  MPG_ARG(n, "mpg_device_count: NULL argument");
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
  *n = c;
  return MPG_SUCCESS;
}

int mpg_device_info(char *arch_buf, int buf_len, int *n_cu, int64_t *hbm_bytes) {
  MPG_CHECK_INIT();
  hipDeviceProp_t p;
  MPG_HIP(hipGetDeviceProperties(&p, g_device));
  if (arch_buf && buf_len > 0) {
    strncpy(arch_buf, p.gcnArchName, (size_t)buf_len - 1);
    arch_buf[buf_len - 1] = 0;
  }
  if (n_cu) *n_cu = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
  return MPG_SUCCESS;
}

// ---- mesh -----------------------------------------------------------------------------------------
static std::vector<mpg_mesh_s *> g_meshes;   // live meshes: a grid that goes away tells the meshes cut to it (geo_grid)

static int mesh_common_checks(int64_t nCells, int64_t nVertices, int maxEdges, const double *latCell, const double *lonCell, const double *latVertex,
                              const double *lonVertex, const int32_t *verticesOnCell, mpg_mesh *out, const char *who) {
  if (!out) { mpg_set_error("%s: out is NULL", who); return MPG_ERR_INVALID_ARG; }
  if (!(nCells > 0 && nVertices > 0 && maxEdges >= 3)) { mpg_set_error("%s: nCells, nVertices must be > 0 and maxEdges >= 3", who); return MPG_ERR_INVALID_ARG; }
  if (!(nCells < 0x7fffffff && nVertices < 0x7fffffff)) { mpg_set_error("%s: sizes must fit int32 ids", who); return MPG_ERR_INVALID_ARG; }
  if (!(latCell && lonCell && latVertex && lonVertex && verticesOnCell)) { mpg_set_error("%s: NULL array", who); return MPG_ERR_INVALID_ARG; }
  return MPG_SUCCESS;
}

// ESMF_MeshCreate for ONE rank of a row-sharded job (include/mpassit_amd.h; k_mesh_window.hip has the method and the argument
// why the weights equal those of the whole mesh)
int mpg_mesh_create_window(int64_t nCells, int64_t nVertices, int maxEdges, const double *latCell, const double *lonCell, const double *latVertex,
                           const double *lonVertex, const int32_t *verticesOnCell, mpg_grid grid, mpg_mesh *out) {
  MPG_CHECK_INIT();
  int rc = mesh_common_checks(nCells, nVertices, maxEdges, latCell, lonCell, latVertex, lonVertex, verticesOnCell, out, "mpg_mesh_create_window");
  if (rc) return rc;
  MPG_ARG(grid, "mpg_mesh_create_window: NULL grid");
  mpg_mesh_s *m = new mpg_mesh_s();
  m->nCells = nCells;
  m->nVertices = nVertices;
  m->maxEdges = maxEdges;
  m->geo_grid = grid;
  g_meshes.push_back(m);
  if ((rc = mpg_k_mesh_coords(nCells, lonCell, latCell, m->cell, g_stream)) ||
      (rc = mpg_k_mesh_window(m, grid, latVertex, lonVertex, verticesOnCell, g_stream))) {
    mpg_mesh_destroy(m);
    return rc;
  }
  *out = m;
  return MPG_SUCCESS;
}

int mpg_mesh_window_info(mpg_mesh m, int64_t *cell_first, int64_t *cell_count, int64_t *vertex_first, int64_t *vertex_count, double *margin) {
  MPG_ARG(m, "mpg_mesh_window_info: NULL mesh");
  if (cell_first) *cell_first = m->cw0;
  if (cell_count) *cell_count = m->cwn;
  if (vertex_first) *vertex_first = m->vw0;
  if (vertex_count) *vertex_count = m->vwn;
  if (margin) *margin = m->geo_margin;
  return MPG_SUCCESS;
}

int mpg_mesh_create(int64_t nCells, int64_t nVertices, int maxEdges, const double *latCell, const double *lonCell,
                    const double *latVertex, const double *lonVertex, const int32_t *verticesOnCell, mpg_mesh *out) {
  MPG_CHECK_INIT();
  {
    int rc0 = mesh_common_checks(nCells, nVertices, maxEdges, latCell, lonCell, latVertex, lonVertex, verticesOnCell, out, "mpg_mesh_create");
    if (rc0) return rc0;
  }
  mpg_mesh_s *m = new mpg_mesh_s();
  m->nCells = nCells;
  m->nVertices = nVertices;
  m->maxEdges = maxEdges;
  m->cwn = nCells;      // geometry window = the whole mesh
  m->vwn = nVertices;
  m->geo_margin = 4.0;
  g_meshes.push_back(m);
  int rc;
  hipStream_t s = g_stream;
  if ((rc = mpg_k_mesh_coords(nCells, lonCell, latCell, m->cell, s)) || (rc = mpg_k_mesh_coords(nVertices, lonVertex, latVertex, m->vert, s)) ||
      (rc = m->voc.alloc((size_t)nCells * maxEdges))) {
    mpg_mesh_destroy(m);
    return rc;
  }
  hipError_t he = hipMemcpyAsync(m->voc.p, verticesOnCell, sizeof(int32_t) * (size_t)nCells * maxEdges, hipMemcpyHostToDevice, s);
  if (he != hipSuccess) {
    mpg_set_error("mpg_mesh_create: uploading verticesOnCell failed: %s", hipGetErrorString(he));
    mpg_mesh_destroy(m);
    return MPG_ERR_HIP;
  }
  if ((rc = mpg_k_voc_check(m->voc.p, nCells * maxEdges, nVertices, "mpg_mesh_create", s)) || (rc = mpg_k_dual_triangles(m, s))) {
    mpg_mesh_destroy(m);
    return rc;
  }
  *out = m;
  return MPG_SUCCESS;
}

// handles outlive neither their mesh nor their grid in the cache: a recycled address must never hit
static void cache_purge(void *obj) {
  store_worker_drain();   // a Store in flight reads the mesh and the grid it was started on
  CACHE_LOCK();
  drop_parked(obj);
  for (auto it = g_cache.begin(); it != g_cache.end();) {
    if (std::get<0>(it->first) == obj || std::get<2>(it->first) == obj) {
      it->second->cached = false;
      it = g_cache.erase(it);
    } else {
      ++it;
    }
  }
}

int mpg_mesh_destroy(mpg_mesh m) {
  if (!m) return MPG_SUCCESS;
  cache_purge(m);
  for (size_t i = 0; i < g_meshes.size(); ++i)
    if (g_meshes[i] == m) {
      g_meshes.erase(g_meshes.begin() + (long)i);
      break;
    }
  m->cell.free();
  m->vert.free();
  m->voc.free();
  m->tri.free();
  m->fan.free();
  m->bvh.free();
  delete m;
  return MPG_SUCCESS;
}

int mpg_mesh_get_triangles(mpg_mesh m, int32_t *tri_host) {
  MPG_CHECK_INIT();
  MPG_ARG(m && tri_host, "mpg_mesh_get_triangles: NULL argument");
  const int64_t nV = m->vwn;   // vertices outside the geometry window of a mesh cut to a grid have no triangle here
  for (int64_t i = 0; i < 3 * m->nVertices; ++i) tri_host[i] = -1;
  if (nV == 0) return MPG_SUCCESS;
  std::vector<int32_t> tmp(3 * (size_t)nV);
  MPG_HIP(hipMemcpy(tmp.data(), m->tri.p, sizeof(int32_t) * 3 * nV, hipMemcpyDeviceToHost));
  for (int64_t v = 0; v < nV; ++v)
    for (int k = 0; k < 3; ++k) tri_host[3 * (m->vw0 + v) + k] = tmp[(size_t)k * nV + v];
  return MPG_SUCCESS;
}

// ---- grid -----------------------------------------------------------------------------------------
int mpg_grid_create(int nx, int ny, int periodic_i, const double *lon_center, const double *lat_center, const double *lon_corner,
                    const double *lat_corner, const double *lon_edge1, const double *lat_edge1, const double *lon_edge2,
                    const double *lat_edge2, mpg_grid *out) {
  MPG_CHECK_INIT();
  MPG_ARG(out, "mpg_grid_create: out is NULL");
  MPG_ARG(nx > 0 && ny > 0, "mpg_grid_create: nx, ny must be > 0");
  MPG_ARG(lon_center && lat_center, "mpg_grid_create: CENTER coordinates are mandatory");
  MPG_ARG(periodic_i >= 0 && periodic_i < 8 && ((periodic_i & MPG_GRID_PERIODIC_I) || !periodic_i),
          "mpg_grid_create: periodic_i must be 0 or MPG_GRID_PERIODIC_I [| MPG_GRID_NO_SOUTH_POLE | MPG_GRID_NO_NORTH_POLE]");
  MPG_ARG((int64_t)(nx + 1) * (ny + 1) < 0x7fffffff, "mpg_grid_create: grid too large for int32 ids");
  mpg_grid_s *g = new mpg_grid_s();
  g->nx = nx;
  g->ny = ny;
  g->periodic = periodic_i;
  g->snx[MPG_STAGGERLOC_CENTER] = nx;     g->sny[MPG_STAGGERLOC_CENTER] = ny;
  g->snx[MPG_STAGGERLOC_EDGE1] = nx + 1;  g->sny[MPG_STAGGERLOC_EDGE1] = ny;
  g->snx[MPG_STAGGERLOC_EDGE2] = nx;      g->sny[MPG_STAGGERLOC_EDGE2] = ny + 1;
  g->snx[MPG_STAGGERLOC_CORNER] = nx + 1; g->sny[MPG_STAGGERLOC_CORNER] = ny + 1;
  const double *lon[4] = {lon_center, lon_edge1, lon_edge2, lon_corner};
  const double *lat[4] = {lat_center, lat_edge1, lat_edge2, lat_corner};
  for (int st = 0; st < 4; ++st) {
    if (!lon[st] || !lat[st]) continue;
    int rc = mpg_k_grid_coords((int64_t)g->snx[st] * g->sny[st], lon[st], lat[st], g->pts[st], g_stream);
    if (rc) {
      mpg_grid_destroy(g);
      return rc;
    }
  }
  *out = g;
  return MPG_SUCCESS;
}

int mpg_grid_destroy(mpg_grid g) {
  if (!g) return MPG_SUCCESS;
  cache_purge(g);
  for (mpg_mesh_s *m : g_meshes)   // a mesh cut to this grid serves no other: a recycled address must never pass for it
    if (m->geo_grid == g) m->geo_grid_gone = true;
  for (int st = 0; st < 4; ++st) {
    g->pts[st].free();
    g->pyr[st].free();
  }
  g->cellpyr.free();
  for (int st = 0; st < 4; ++st) {
    g->lon[st].free();
    g->lat[st].free();
  }
  for (int st = 0; st < 3; ++st) g->mapfac[st].free();
  g->cosa.free();
  g->sina.free();
  delete g;
  return MPG_SUCCESS;
}

static void grid_shape(mpg_grid_s *g, int nx, int ny, int periodic_i) {
  g->nx = nx;
  g->ny = ny;
  g->periodic = periodic_i;
  g->snx[MPG_STAGGERLOC_CENTER] = nx;     g->sny[MPG_STAGGERLOC_CENTER] = ny;
  g->snx[MPG_STAGGERLOC_EDGE1] = nx + 1;  g->sny[MPG_STAGGERLOC_EDGE1] = ny;
  g->snx[MPG_STAGGERLOC_EDGE2] = nx;      g->sny[MPG_STAGGERLOC_EDGE2] = ny + 1;
  g->snx[MPG_STAGGERLOC_CORNER] = nx + 1; g->sny[MPG_STAGGERLOC_CORNER] = ny + 1;
}

int mpg_grid_create_proj(const mpg_proj *proj, int nx, int ny, int periodic_i, mpg_grid *out) {
  MPG_CHECK_INIT();
  MPG_ARG(proj && out, "mpg_grid_create_proj: NULL argument");
  MPG_ARG(nx > 0 && ny > 0, "mpg_grid_create_proj: nx, ny must be > 0");
  MPG_ARG((int64_t)(nx + 1) * (ny + 1) < 0x7fffffff, "mpg_grid_create_proj: grid too large for int32 ids");
  MPG_ARG(periodic_i >= 0 && periodic_i < 8 && ((periodic_i & MPG_GRID_PERIODIC_I) || !periodic_i),
          "mpg_grid_create_proj: periodic_i must be 0 or MPG_GRID_PERIODIC_I [| MPG_GRID_NO_SOUTH_POLE | MPG_GRID_NO_NORTH_POLE]");
  mpg_grid_s *g = new mpg_grid_s();
  grid_shape(g, nx, ny, periodic_i);
  int rc = mpg_k_target_grid(proj, g, g_stream);
  if (rc) {
    mpg_grid_destroy(g);
    return rc;
  }
  *out = g;
  return MPG_SUCCESS;
}

int mpg_grid_attach_proj(mpg_grid g, const mpg_proj *proj, int row0) {
  MPG_CHECK_INIT();
  MPG_ARG(g && proj, "mpg_grid_attach_proj: NULL argument");
  MPG_ARG(row0 >= 0, "mpg_grid_attach_proj: row0 must be >= 0");
  return mpg_k_attach_proj(g, proj, row0, g_stream);
}

#define MPG_PROJ_GRID(g, what)                                                                  \
  do {                                                                                          \
    MPG_ARG(g, what ": NULL grid");                                                             \
    if (!(g)->from_proj) {                                                                      \
      mpg_set_error(what ": the grid was created from caller arrays (mpg_grid_create), which the caller still holds"); \
      return MPG_ERR_UNSUPPORTED;                                                               \
    }                                                                                           \
  } while (0)

int mpg_grid_get_coords(mpg_grid g, int staggerloc, double *lon_host, double *lat_host) {
  MPG_CHECK_INIT();
  MPG_PROJ_GRID(g, "mpg_grid_get_coords");
  MPG_ARG(staggerloc >= 0 && staggerloc < 4, "mpg_grid_get_coords: bad staggerloc");
  size_t bytes = sizeof(double) * (size_t)g->snx[staggerloc] * g->sny[staggerloc];
  if (lon_host) MPG_HIP(hipMemcpy(lon_host, g->lon[staggerloc].p, bytes, hipMemcpyDeviceToHost));
  if (lat_host) MPG_HIP(hipMemcpy(lat_host, g->lat[staggerloc].p, bytes, hipMemcpyDeviceToHost));
  return MPG_SUCCESS;
}

int mpg_grid_get_mapfac(mpg_grid g, int staggerloc, double *mapfac_host) {
  MPG_CHECK_INIT();
  MPG_PROJ_GRID(g, "mpg_grid_get_mapfac");
  MPG_ARG(staggerloc >= 0 && staggerloc < 3 && mapfac_host, "mpg_grid_get_mapfac: CENTER, EDGE1 or EDGE2 and a buffer");
  MPG_HIP(hipMemcpy(mapfac_host, g->mapfac[staggerloc].p, sizeof(double) * (size_t)g->snx[staggerloc] * g->sny[staggerloc],
                    hipMemcpyDeviceToHost));
  return MPG_SUCCESS;
}

int mpg_grid_rotang_dev(mpg_grid g, const double **cosa_dev, const double **sina_dev) {
  MPG_CHECK_INIT();
  MPG_PROJ_GRID(g, "mpg_grid_rotang_dev");
  if (!g->cosa.p) {
    mpg_set_error("mpg_grid_rotang_dev: cos/sin(alpha) exist for PROJ_LC grids only (model_grid.F90:1113)");
    return MPG_ERR_UNSUPPORTED;
  }
  if (cosa_dev) *cosa_dev = g->cosa.p;
  if (sina_dev) *sina_dev = g->sina.p;
  return MPG_SUCCESS;
}

int mpg_grid_get_rotang(mpg_grid g, double *cosa_host, double *sina_host) {
  const double *c, *s;
  int rc = mpg_grid_rotang_dev(g, &c, &s);
  if (rc) return rc;
  size_t bytes = sizeof(double) * (size_t)g->nx * g->ny;
  if (cosa_host) MPG_HIP(hipMemcpy(cosa_host, c, bytes, hipMemcpyDeviceToHost));
  if (sina_host) MPG_HIP(hipMemcpy(sina_host, s, bytes, hipMemcpyDeviceToHost));
  return MPG_SUCCESS;
}

// ---- RegridStore ----------------------------------------------------------------------------------
static void handle_free(mpg_handle_s *h) {
  h->idx.free();
  h->w.free();
  h->rowptr.free();
  h->col.free();
  h->val.free();
  h->pole_dst.free();
  h->pole_src0.free();
  h->pole_w.free();
  h->free_tile_lists();
  delete h;
}

struct StoreCtx {
  mpg_mesh_s *m;
  mpg_grid_s *g;
  int stagger;
  int method;
  int meshloc;
};

// ---- the Store worker -------------------------------------------------------------------------------------------------
// interp.F90:207-437 stores its weight sets one after the other, each in front of the Regrids that use it; they are
// independent of each other and of every Regrid that does not use them.  All RegridStores of this library run on ONE worker
// thread with a stream of its own: mpg_regrid_store[_grid]_begin queues a Store and returns; mpg_regrid_store[_grid] returns
// the finished handle of its key -- from the cache, from the queue (waiting for it), or by queueing it itself and waiting.
// While the worker builds the conservative and nearest-neighbour weights (3 of the 4.6 ms of configuration 4's Stores,
// compute-bound kernels) the caller's thread is already issuing the bilinear Regrids (HBM-bound) on ITS stream.  One
// Store at a time: the lazily built search structures of a mesh / grid (pyramids, BVH, fans) are only ever touched by this
// thread.  The weights do not depend on which thread builds them.
struct StoreJob {
  HandleKey key;
  int (*build)(mpg_handle_s *, void *, hipStream_t);
  StoreCtx ctx;
  hipEvent_t ready = nullptr;   // recorded on the set-up stream when the job was queued: the mesh / grid uploads and kernels issued there come first
  bool done = false;
  int rc = MPG_SUCCESS;
  char err[1024] = "";
  ~StoreJob() {
    if (ready) (void)hipEventDestroy(ready);
  }
};
// The worker's state lives on the heap and is never destroyed: a process that ends without mpg_finalize (a Fortran `stop`, an error exit)
// must not run the destructor of a condition variable the worker is waiting on (that blocks for ever).
struct StoreWorker {
  std::mutex mu, start_mu;
  std::condition_variable cv;
  std::deque<std::shared_ptr<StoreJob>> queue;
  std::map<HandleKey, std::shared_ptr<StoreJob>> pending;   // queued or running
  std::thread *thread = nullptr;
  hipStream_t stream = nullptr;
  bool stop = false;
};
static StoreWorker &SW() {
  static StoreWorker *w = new StoreWorker();
  return *w;
}
static const char *mpg_last_error_cstr() { return g_err; }

static int store_build_now(StoreJob &job, hipStream_t s) {
  mpg_handle_s *h = new mpg_handle_s();
  h->refcount = 0;   // nobody holds it yet: parked in the cache until a mpg_regrid_store collects it
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    mpg_set_error("hipEventCreate failed in a RegridStore");
    handle_free(h);
    return MPG_ERR_HIP;
  }
  (void)hipEventRecord(e0, s);
  int rc = job.build(h, &job.ctx, s);
  if (!rc) {
    hipError_t e = hipEventRecord(e1, s);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&h->store_ms, e0, e1);
    if (e != hipSuccess) {
      mpg_set_error("RegridStore: %s", hipGetErrorString(e));
      rc = MPG_ERR_HIP;
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc) {
    handle_free(h);
    return rc;
  }
  CACHE_LOCK();
  h->key = job.key;
  h->cached = true;
  h->parked_at = ++g_park_clock;
  g_cache[job.key] = h;
  return MPG_SUCCESS;
}

static void store_worker_main(int device) {
  // a worker that cannot bind its device keeps serving the queue -- with the error: a thread that simply left would make every later
  // RegridStore wait for ever on a job nobody runs
  const hipError_t dev_err = hipSetDevice(device);
  for (;;) {
    std::shared_ptr<StoreJob> job;
    {
      std::unique_lock<std::mutex> lk(SW().mu);
      SW().cv.wait(lk, [] { return SW().stop || !SW().queue.empty(); });
      if (SW().queue.empty()) return;   // stop asked and nothing left
      job = SW().queue.front();
      SW().queue.pop_front();
    }
    g_err[0] = 0;
    int rc;
    if (dev_err != hipSuccess) {
      mpg_set_error("RegridStore: the Store worker could not bind device %d: %s", device, hipGetErrorString(dev_err));
      rc = MPG_ERR_HIP;
    } else {
      if (job->ready) (void)hipStreamWaitEvent(SW().stream, job->ready, 0);
      rc = store_build_now(*job, SW().stream);
    }
    {
      std::lock_guard<std::mutex> lk(SW().mu);
      job->rc = rc;
      if (rc) snprintf(job->err, sizeof(job->err), "%s", mpg_last_error_cstr());
      job->done = true;
      SW().pending.erase(job->key);
    }
    SW().cv.notify_all();
  }
}

static int store_worker_start(int device) {   // mpg_init's helper thread starts it; a Store that comes first starts it itself
  std::lock_guard<std::mutex> lk(SW().start_mu);
  if (SW().thread) return MPG_SUCCESS;
  MPG_HIP(hipStreamCreateWithFlags(&SW().stream, hipStreamNonBlocking));
  SW().stop = false;
  SW().thread = new std::thread(store_worker_main, device);
  return MPG_SUCCESS;
}

static void store_worker_drain() {
  std::unique_lock<std::mutex> lk(SW().mu);
  SW().cv.wait(lk, [] { return SW().pending.empty(); });
}

static void store_worker_stop() {   // mpg_finalize
  if (!SW().thread) return;
  store_worker_drain();
  {
    std::lock_guard<std::mutex> lk(SW().mu);
    SW().stop = true;
  }
  SW().cv.notify_all();
  SW().thread->join();
  delete SW().thread;
  SW().thread = nullptr;
  (void)hipStreamDestroy(SW().stream);
  SW().stream = nullptr;
}

// out == nullptr: start the Store and return (mpg_regrid_store[_grid]_begin); else return the handle, waiting for its Store
static int store_common(HandleKey key, mpg_handle *out, int (*build)(mpg_handle_s *, void *, hipStream_t), const StoreCtx &ctx) {
  for (int attempt = 0; attempt < 2; ++attempt) {
    {
      CACHE_LOCK();
      auto it = g_cache.find(key);
      if (it != g_cache.end()) {   // in use elsewhere, or parked by an earlier release / a finished _begin: the same weights, no device work
        if (out) {
          it->second->refcount++;
          *out = it->second;
        }
        return MPG_SUCCESS;
      }
    }
    int rc = store_worker_start(g_device);
    if (rc) return rc;
    std::shared_ptr<StoreJob> job;
    {
      std::unique_lock<std::mutex> lk(SW().mu);
      auto pit = SW().pending.find(key);
      if (pit != SW().pending.end()) {
        job = pit->second;
      } else {
        job = std::make_shared<StoreJob>();
        job->key = key;
        job->build = build;
        job->ctx = ctx;
        if (hipEventCreateWithFlags(&job->ready, hipEventDisableTiming) == hipSuccess) (void)hipEventRecord(job->ready, g_stream);
        SW().pending[key] = job;
        SW().queue.push_back(job);
      }
      if (!out) {
        lk.unlock();
        SW().cv.notify_all();
        return MPG_SUCCESS;
      }
      SW().cv.notify_all();
      SW().cv.wait(lk, [&] { return job->done; });
    }
    if (job->rc) {
      mpg_set_error("%s", job->err);
      return job->rc;
    }
    // finished: the handle is in the cache now (unless a flood of releases has pushed it out again: then once more)
  }
  mpg_set_error("RegridStore: the finished handle left the cache before it was collected");
  return MPG_ERR_HIP;
}

static int store_mesh_build(mpg_handle_s *h, void *c, hipStream_t s) {
  StoreCtx *x = (StoreCtx *)c;
  h->method = x->method;
  int rc;
  if (x->method == MPG_REGRIDMETHOD_BILINEAR) rc = mpg_k_store_bilinear_mesh(x->m, x->g, x->stagger, x->meshloc, h, s);
  else if (x->method == MPG_REGRIDMETHOD_NEAREST_STOD) rc = mpg_k_store_nearest(x->m, x->g, x->stagger, h, s);
  else rc = mpg_k_store_conserve(x->m, x->g, h, s);
  if (!rc && x->m->win_count[x->meshloc] >= 0)    // the mesh's sources are windowed: index relative to the window from the start
    rc = mpg_k_rebase(h, x->m->win_first[x->meshloc], x->m->win_count[x->meshloc], s, true);
  return rc;
}
static int store_grid_build(mpg_handle_s *h, void *c, hipStream_t s) {
  StoreCtx *x = (StoreCtx *)c;
  h->method = x->method;
  return mpg_k_store_grid_bilinear(x->g, x->stagger, h, s);
}

static int store_mesh_entry(mpg_mesh src, int src_meshloc, mpg_grid dst, int dst_staggerloc, int regridmethod, mpg_handle *out, const char *who) {
  MPG_CHECK_INIT();
  MPG_ARG(src && dst, "mpg_regrid_store: NULL argument");
  MPG_ARG(dst_staggerloc >= 0 && dst_staggerloc <= 2, "mpg_regrid_store: destination stagger must be CENTER, EDGE1 or EDGE2");
  MPG_ARG(src_meshloc == MPG_MESHLOC_ELEMENT || src_meshloc == MPG_MESHLOC_NODE, "mpg_regrid_store: unknown mesh location");
  if (src_meshloc == MPG_MESHLOC_NODE && regridmethod != MPG_REGRIDMETHOD_BILINEAR) {
    mpg_set_error("%s: node-located sources are only regridded bilinearly by the reference (interp.F90:350-366)", who);
    return MPG_ERR_UNSUPPORTED;
  }
  MPG_ARG(regridmethod >= 0 && regridmethod <= 2, "mpg_regrid_store: unknown regrid method");
  if (src->geo_grid && (src->geo_grid != dst || src->geo_grid_gone)) {
    mpg_set_error("%s: the mesh was cut to another grid (mpg_mesh_create_window%s); it holds only the cells that grid can see", who,
                  src->geo_grid_gone ? ", which has been destroyed" : "");
    return MPG_ERR_INVALID_ARG;
  }
  if (regridmethod == MPG_REGRIDMETHOD_CONSERVE && dst_staggerloc != MPG_STAGGERLOC_CENTER) {
    mpg_set_error("%s: conservative regridding is defined on the CENTER stagger only", who);
    return MPG_ERR_UNSUPPORTED;
  }
  StoreCtx ctx{src, dst, dst_staggerloc, regridmethod, src_meshloc};
  // the line type in force is part of what a bilinear handle IS: the two never share a cache entry
  // ... and so is the apex rule of the polygon fans for node-located sources
  HandleKey key(src, src_meshloc, dst, dst_staggerloc,
                regridmethod + (regridmethod == MPG_REGRIDMETHOD_BILINEAR ? 16 * mpg_bilinear_linetype() : 0) +
                    (regridmethod == MPG_REGRIDMETHOD_BILINEAR && src_meshloc == MPG_MESHLOC_NODE ? 256 * (mpg_node_fan_origin() + 16) : 0));
  return store_common(key, out, store_mesh_build, ctx);
}

int mpg_regrid_store(mpg_mesh src, int src_meshloc, mpg_grid dst, int dst_staggerloc, int regridmethod, mpg_handle *out) {
  MPG_CHECK_INIT();
  MPG_ARG(out, "mpg_regrid_store: NULL argument");
  return store_mesh_entry(src, src_meshloc, dst, dst_staggerloc, regridmethod, out, "mpg_regrid_store");
}
int mpg_regrid_store_begin(mpg_mesh src, int src_meshloc, mpg_grid dst, int dst_staggerloc, int regridmethod) {
  return store_mesh_entry(src, src_meshloc, dst, dst_staggerloc, regridmethod, nullptr, "mpg_regrid_store_begin");
}

// ---- source window of a mesh (a host that holds only the cells its target rows reference) ---------------------------------
int mpg_handle_source_range(mpg_handle h, int64_t *first, int64_t *end) {
  MPG_CHECK_INIT();
  MPG_ARG(h && first && end, "mpg_handle_source_range: NULL argument");
  MPG_ARG(!h->localized, "mpg_handle_source_range: the handle was re-indexed (mpg_handle_localize / mpg_handle_rebase)");
  int rc = mpg_k_source_range(h, first, end, g_stream);
  if (rc) return rc;
  if (h->cached && std::get<1>(h->key) < 100) {   // Mesh -> Grid handle of a windowed mesh: back to global ids
    mpg_mesh_s *m = (mpg_mesh_s *)std::get<0>(h->key);
    const int loc = std::get<1>(h->key);
    if (m->win_count[loc] >= 0 && *end > *first) {
      *first += m->win_first[loc];
      *end += m->win_first[loc];
    }
  }
  return MPG_SUCCESS;
}

int mpg_mesh_set_source_window(mpg_mesh m, int meshloc, int64_t first, int64_t count) {
  MPG_CHECK_INIT();
  MPG_ARG(m && (meshloc == MPG_MESHLOC_ELEMENT || meshloc == MPG_MESHLOC_NODE), "mpg_mesh_set_source_window: bad mesh / location");
  const int64_t n_all = meshloc == MPG_MESHLOC_ELEMENT ? m->nCells : m->nVertices;
  MPG_ARG(first >= 0 && count >= 0 && first + count <= n_all, "mpg_mesh_set_source_window: the window must lie inside the mesh");
  const int64_t old_first = m->win_count[meshloc] >= 0 ? m->win_first[meshloc] : 0;
  store_worker_drain();
  CACHE_LOCK();
  struct Busy { Busy() { g_cache_busy = true; } ~Busy() { g_cache_busy = false; } } busy;
  // every handle of this mesh and location that exists already (in use or parked) moves to the new window -- after ALL of
  // them have been checked: one that references a source outside it fails the call and nothing has changed
  for (auto &kv : g_cache) {
    if (std::get<0>(kv.first) != (void *)m || std::get<1>(kv.first) != meshloc) continue;
    int64_t f = 0, e = 0;
    int rc = mpg_k_source_range(kv.second, &f, &e, g_stream);
    if (rc) return rc;
    if (e > f && (f + old_first < first || e + old_first > first + count)) {
      mpg_set_error("mpg_mesh_set_source_window: a route handle of this mesh references sources [%lld, %lld), outside the window [%lld, %lld)",
                    (long long)(f + old_first), (long long)(e + old_first), (long long)first, (long long)(first + count));
      return MPG_ERR_INVALID_ARG;
    }
  }
  for (auto &kv : g_cache) {
    if (std::get<0>(kv.first) != (void *)m || std::get<1>(kv.first) != meshloc) continue;
    mpg_handle_s *h = kv.second;
    h->free_tile_lists();
    h->lf_choice = 0;
    h->cf_choice = 0;
    int rc = mpg_k_rebase(h, first - old_first, count, g_stream, true);
    if (rc) return rc;
  }
  m->win_first[meshloc] = first;
  m->win_count[meshloc] = first == 0 && count == n_all ? -1 : count;   // the whole mesh: no window any more
  return MPG_SUCCESS;
}

static int store_grid_entry(mpg_grid grid, int src_staggerloc, int dst_staggerloc, int regridmethod, mpg_handle *out) {
  MPG_CHECK_INIT();
  MPG_ARG(grid, "mpg_regrid_store_grid: NULL argument");
  if (src_staggerloc != MPG_STAGGERLOC_CENTER || (dst_staggerloc != MPG_STAGGERLOC_EDGE1 && dst_staggerloc != MPG_STAGGERLOC_EDGE2) ||
      regridmethod != MPG_REGRIDMETHOD_BILINEAR) {
    mpg_set_error("mpg_regrid_store_grid: only bilinear CENTER -> EDGE1/EDGE2 is used by the reference (interp.F90:298,316)");
    return MPG_ERR_UNSUPPORTED;
  }
  StoreCtx ctx{nullptr, grid, dst_staggerloc, regridmethod, 0};
  HandleKey key(grid, 100 + src_staggerloc, grid, dst_staggerloc, regridmethod + 256 * mpg_grid_inside_tol_exp());   // the inside tolerance in force is part of the handle
  return store_common(key, out, store_grid_build, ctx);
}
int mpg_regrid_store_grid(mpg_grid grid, int src_staggerloc, int dst_staggerloc, int regridmethod, mpg_handle *out) {
  MPG_CHECK_INIT();
  MPG_ARG(out, "mpg_regrid_store_grid: NULL argument");
  return store_grid_entry(grid, src_staggerloc, dst_staggerloc, regridmethod, out);
}
int mpg_regrid_store_grid_begin(mpg_grid grid, int src_staggerloc, int dst_staggerloc, int regridmethod) {
  return store_grid_entry(grid, src_staggerloc, dst_staggerloc, regridmethod, nullptr);
}

int mpg_handle_release(mpg_handle h) {
  if (!h) return MPG_SUCCESS;
  CACHE_LOCK();
  if (--h->refcount > 0) return MPG_SUCCESS;
  if (!h->cached) {
    handle_free(h);
    return MPG_SUCCESS;
  }
  h->parked_at = ++g_park_clock;   // stays in the cache with refcount 0
  int nparked = 0;
  for (auto &kv : g_cache) nparked += kv.second->refcount == 0;
  while (nparked > MPG_MAX_PARKED) {   // drop the one released longest ago
    auto old = g_cache.end();
    for (auto it = g_cache.begin(); it != g_cache.end(); ++it)
      if (it->second->refcount == 0 && (old == g_cache.end() || it->second->parked_at < old->second->parked_at)) old = it;
    mpg_handle_s *victim = old->second;
    g_cache.erase(old);
    handle_free(victim);
    --nparked;
  }
  return MPG_SUCCESS;
}

// ---- Regrid ---------------------------------------------------------------------------------------
int mpg_regrid_dev(mpg_handle h, const double *src_dev, int src_layout, int nlev, int nfields, double *dst_dev, void *hip_stream) {
  MPG_CHECK_INIT();
  MPG_ARG(h && dst_dev && (src_dev || h->n_src == 0), "mpg_regrid: NULL argument");
  MPG_ARG(nlev >= 1 && nfields >= 1, "mpg_regrid: nlev and nfields must be >= 1");
  MPG_ARG(src_layout == MPG_LAYOUT_CELL_FAST || src_layout == MPG_LAYOUT_LEV_FAST, "mpg_regrid: bad src_layout");
  return mpg_k_apply(h, src_dev, src_layout, nlev, nfields, dst_dev, (hipStream_t)hip_stream);
}

int mpg_regrid_typed_dev(mpg_handle h, const void *src_dev, int src_type, int src_layout, int nlev, int nfields, void *dst_dev,
                         int dst_type, double scale, double offset, void *hip_stream) {
  MPG_CHECK_INIT();
  MPG_ARG(h && dst_dev && (src_dev || h->n_src == 0), "mpg_regrid_typed: NULL argument");
  MPG_ARG(nlev >= 1 && nfields >= 1, "mpg_regrid_typed: nlev and nfields must be >= 1");
  MPG_ARG(src_layout == MPG_LAYOUT_CELL_FAST || src_layout == MPG_LAYOUT_LEV_FAST, "mpg_regrid_typed: bad src_layout");
  MPG_ARG(src_type >= 0 && src_type <= 3 && dst_type >= 0 && dst_type <= 3, "mpg_regrid_typed: src_type / dst_type must be MPG_TYPE_F64 or MPG_TYPE_F32, optionally | MPG_TYPE_BE");
  return mpg_k_apply_typed(h, src_dev, src_type, src_layout, nlev, nfields, dst_dev, dst_type, scale, offset, (hipStream_t)hip_stream);
}

int mpg_regrid_bundle_typed_dev(mpg_handle h, int nfields, const void *const *src_dev, int src_type, int src_layout, int nlev,
                                void *const *dst_dev, int dst_type, double scale, const double *offsets, void *hip_stream) {
  MPG_CHECK_INIT();
  MPG_ARG(h && src_dev && dst_dev, "mpg_regrid_bundle_typed: NULL argument");
  MPG_ARG(nlev >= 1 && nfields >= 1, "mpg_regrid_bundle_typed: nlev and nfields must be >= 1");
  MPG_ARG(src_layout == MPG_LAYOUT_CELL_FAST || src_layout == MPG_LAYOUT_LEV_FAST, "mpg_regrid_bundle_typed: bad src_layout");
  MPG_ARG(src_type >= 0 && src_type <= 3 && dst_type >= 0 && dst_type <= 3, "mpg_regrid_bundle_typed: src_type / dst_type must be MPG_TYPE_F64 or MPG_TYPE_F32, optionally | MPG_TYPE_BE");
  for (int f = 0; f < nfields; ++f) MPG_ARG(dst_dev[f] && (src_dev[f] || h->n_src == 0), "mpg_regrid_bundle_typed: NULL field pointer");
  hipStream_t s = (hipStream_t)hip_stream;
  if (nfields == 1 || h->n_src == 0 || h->n_dst == 0) {   // nothing to share between fields: the plain call, field by field
    for (int f = 0; f < nfields; ++f) {
      int rc = mpg_k_apply_typed(h, src_dev[f], src_type, src_layout, nlev, 1, dst_dev[f], dst_type, scale, offsets ? offsets[f] : 0.0, s);
      if (rc) return rc;
    }
    return MPG_SUCCESS;
  }
  for (int f0 = 0; f0 < nfields; f0 += MPG_TAB_MAX) {   // the pointers travel in the kernels' argument blocks, MPG_TAB_MAX fields per launch
    FieldTab tab;
    tab.n = std::min(MPG_TAB_MAX, nfields - f0);
    for (int k = 0; k < tab.n; ++k) {
      tab.src[k] = src_dev[f0 + k];
      tab.dst[k] = dst_dev[f0 + k];
      tab.off[k] = offsets ? offsets[f0 + k] : 0.0;
    }
    int rc = mpg_k_apply_typed(h, nullptr, src_type, src_layout, nlev, tab.n, nullptr, dst_type, scale, 0.0, s, tab);
    if (rc) return rc;
  }
  return MPG_SUCCESS;
}

int mpg_regrid(mpg_handle h, const double *src_host, int src_layout, int nlev, int nfields, double *dst_host) {
  MPG_CHECK_INIT();
  MPG_ARG(h && src_host && dst_host, "mpg_regrid: NULL argument");
  MPG_ARG(nlev >= 1 && nfields >= 1, "mpg_regrid: nlev and nfields must be >= 1");
  // float64 in, float64 out through the chunked upload / kernel / download pipeline (mpg_hostpipe.hip); same kernels and
  // the same bits as mpg_regrid_dev
  return mpg_regrid_typed(h, src_host, 0, src_layout, nlev, nfields, dst_host, 0, 1.0, 0.0);
}

int mpg_rotate_winds_dev(int64_t npts, int nlev, const double *cosa_dev, const double *sina_dev, double *u_dev, double *v_dev,
                         void *hip_stream) {
  MPG_CHECK_INIT();
  MPG_ARG(cosa_dev && sina_dev && u_dev && v_dev, "mpg_rotate_winds: NULL argument");
  MPG_ARG(npts >= 0 && nlev >= 1, "mpg_rotate_winds: bad sizes");
  return mpg_k_rotate(npts, nlev, cosa_dev, sina_dev, u_dev, v_dev, (hipStream_t)hip_stream);
}

int mpg_wind_destagger_dev(mpg_handle h1, mpg_handle h2, const double *cosa_dev, const double *sina_dev, const double *umass_dev,
                           const double *vmass_dev, int nlev, void *u_dev, void *v_dev, int dst_type, double *umass_rot_dev, double *vmass_rot_dev,
                           void *hip_stream) {
  MPG_CHECK_INIT();
  MPG_ARG(h1 || h2, "mpg_wind_destagger: no handle");
  MPG_ARG((cosa_dev == nullptr) == (sina_dev == nullptr), "mpg_wind_destagger: cosa and sina come together");
  const bool rot = cosa_dev != nullptr;
  MPG_ARG(!rot || (h1 && h2), "mpg_wind_destagger: the rotation needs both components (interp.F90:291)");
  MPG_ARG((!h1 || u_dev) && (!h2 || v_dev), "mpg_wind_destagger: NULL destination");
  MPG_ARG((!(h1 || rot) || umass_dev) && (!(h2 || rot) || vmass_dev), "mpg_wind_destagger: NULL mass field");
  MPG_ARG(nlev >= 1, "mpg_wind_destagger: nlev must be >= 1");
  MPG_ARG(dst_type >= 0 && dst_type <= 3, "mpg_wind_destagger: dst_type must be MPG_TYPE_F64 or MPG_TYPE_F32, optionally | MPG_TYPE_BE");
  MPG_ARG((!umass_rot_dev || umass_rot_dev != umass_dev) && (!vmass_rot_dev || vmass_rot_dev != vmass_dev),
          "mpg_wind_destagger: the rotated mass winds cannot replace the inputs (neighbouring tiles read them)");
  MPG_ARG(rot || (!umass_rot_dev && !vmass_rot_dev), "mpg_wind_destagger: rotated mass winds asked for without a rotation");
  int rc = mpg_k_wind_destagger(h1, h2, cosa_dev, sina_dev, umass_dev, vmass_dev, nlev, u_dev, v_dev, dst_type, umass_rot_dev, vmass_rot_dev,
                                (hipStream_t)hip_stream);
  if (rc == MPG_ERR_UNSUPPORTED) mpg_set_error("mpg_wind_destagger: the handles are not the CENTER -> EDGE1 / EDGE2 pair of one grid");
  return rc;
}

int mpg_rotate_winds(int64_t npts, int nlev, const double *cosa_host, const double *sina_host, double *u_host, double *v_host) {
  MPG_CHECK_INIT();
  MPG_ARG(cosa_host && sina_host && u_host && v_host, "mpg_rotate_winds: NULL argument");
  MPG_ARG(npts >= 0 && nlev >= 1, "mpg_rotate_winds: bad sizes");
  TmpBuf<double> cs, uv;
  int rc;
  size_t n = (size_t)npts, nl = n * nlev;
  if ((rc = cs.alloc(2 * n)) || (rc = uv.alloc(2 * nl))) {
    cs.free();
    return rc;
  }
  MPG_HIP(hipMemcpyAsync(cs.p, cosa_host, sizeof(double) * n, hipMemcpyHostToDevice, g_stream));
  MPG_HIP(hipMemcpyAsync(cs.p + n, sina_host, sizeof(double) * n, hipMemcpyHostToDevice, g_stream));
  MPG_HIP(hipMemcpyAsync(uv.p, u_host, sizeof(double) * nl, hipMemcpyHostToDevice, g_stream));
  MPG_HIP(hipMemcpyAsync(uv.p + nl, v_host, sizeof(double) * nl, hipMemcpyHostToDevice, g_stream));
  rc = mpg_k_rotate(npts, nlev, cs.p, cs.p + n, uv.p, uv.p + nl, g_stream);
  if (!rc) {
    MPG_HIP(hipMemcpyAsync(u_host, uv.p, sizeof(double) * nl, hipMemcpyDeviceToHost, g_stream));
    MPG_HIP(hipMemcpyAsync(v_host, uv.p + nl, sizeof(double) * nl, hipMemcpyDeviceToHost, g_stream));
    MPG_HIP(hipStreamSynchronize(g_stream));
  }
  cs.free();
  uv.free();
  return rc;
}

// ---- introspection ----------------------------------------------------------------------------------
int mpg_handle_from_weights(int64_t n_src, int nx_dst, int ny_dst, int64_t nnz, const int32_t *row_host, const int32_t *col_host,
                            const double *S_host, mpg_handle *out) {
  MPG_CHECK_INIT();
  MPG_ARG(out && (nnz == 0 || (row_host && col_host && S_host)), "mpg_handle_from_weights: NULL argument");
  MPG_ARG(n_src > 0 && n_src < 0x7fffffff && nx_dst > 0 && ny_dst > 0 && nnz >= 0 && nnz < 0x7fffffff,
          "mpg_handle_from_weights: bad sizes");
  int64_t P = (int64_t)nx_dst * ny_dst;
  std::vector<int32_t> rowptr((size_t)P + 1, 0);
  for (int64_t q = 0; q < nnz; ++q) {
    if (row_host[q] < 1 || row_host[q] > P || col_host[q] < 1 || col_host[q] > n_src) {
      mpg_set_error("mpg_handle_from_weights: entry %lld has row %d / col %d outside [1, n_dst] / [1, n_src]", (long long)q,
                    row_host[q], col_host[q]);
      return MPG_ERR_INVALID_ARG;
    }
    rowptr[(size_t)row_host[q]]++;   // count into slot row (1-based) -> exclusive scan below
  }
  int maxlen = 0, minlen_nonempty = 1 << 30;
  for (int64_t p = 0; p < P; ++p) {
    int len = rowptr[(size_t)p + 1];
    if (len > maxlen) maxlen = len;
    if (len > 0 && len < minlen_nonempty) minlen_nonempty = len;
    rowptr[(size_t)p + 1] += rowptr[(size_t)p];
  }
  std::vector<int32_t> col((size_t)nnz + 1);
  std::vector<double> val((size_t)nnz + 1);
  {
    std::vector<int32_t> cur(rowptr.begin(), rowptr.end() - 1);
    for (int64_t q = 0; q < nnz; ++q) {   // stable: keeps the caller's order inside a row
      int32_t r = row_host[q] - 1;
      col[(size_t)cur[r]] = col_host[q] - 1;
      val[(size_t)cur[r]++] = S_host[q];
    }
  }
  mpg_handle_s *h = new mpg_handle_s();
  h->n_src = n_src;
  h->n_dst = P;
  h->nx_dst = nx_dst;
  h->ny_dst = ny_dst;
  h->nnz = nnz;
  h->method = -1;
  int rc = MPG_SUCCESS;
  if (maxlen == 3 && minlen_nonempty == 3) {
    // every mapped destination has exactly 3 sources (bilinear on triangles): the fast fixed-3 layout
    h->kind = MPG_KIND_FIXED;
    h->nnz_per_row = 3;
    std::vector<int32_t> idx(3 * (size_t)P, -1);
    std::vector<double> w(3 * (size_t)P, 0.0);
    for (int64_t p = 0; p < P; ++p)
      if (rowptr[(size_t)p + 1] > rowptr[(size_t)p])
        for (int k = 0; k < 3; ++k) {
          idx[(size_t)k * P + p] = col[(size_t)rowptr[(size_t)p] + k];
          w[(size_t)k * P + p] = val[(size_t)rowptr[(size_t)p] + k];
        }
    if (!(rc = h->idx.alloc(3 * (size_t)P)) && !(rc = h->w.alloc(3 * (size_t)P))) {
      MPG_HIP(hipMemcpy(h->idx.p, idx.data(), sizeof(int32_t) * 3 * P, hipMemcpyHostToDevice));
      MPG_HIP(hipMemcpy(h->w.p, w.data(), sizeof(double) * 3 * P, hipMemcpyHostToDevice));
    }
  } else {
    h->kind = MPG_KIND_CSR;
    h->nnz_per_row = 0;
    if (!(rc = h->rowptr.alloc((size_t)P + 1)) && !(rc = h->col.alloc((size_t)nnz + 1)) && !(rc = h->val.alloc((size_t)nnz + 1))) {
      MPG_HIP(hipMemcpy(h->rowptr.p, rowptr.data(), sizeof(int32_t) * (P + 1), hipMemcpyHostToDevice));
      MPG_HIP(hipMemcpy(h->col.p, col.data(), sizeof(int32_t) * (nnz + 1), hipMemcpyHostToDevice));
      MPG_HIP(hipMemcpy(h->val.p, val.data(), sizeof(double) * (nnz + 1), hipMemcpyHostToDevice));
    }
  }
  if (rc) {
    handle_free(h);
    return rc;
  }
  *out = h;
  return MPG_SUCCESS;
}

int mpg_handle_info(mpg_handle h, int64_t *n_src, int64_t *n_dst, int *nx_dst, int *ny_dst, int *nnz_per_row, int64_t *nnz) {
  MPG_ARG(h, "mpg_handle_info: NULL handle");
  if (n_src) *n_src = h->n_src;
  if (n_dst) *n_dst = h->n_dst;
  if (nx_dst) *nx_dst = h->nx_dst;
  if (ny_dst) *ny_dst = h->ny_dst;
  if (nnz_per_row) *nnz_per_row = h->nnz_per_row;
  if (nnz) *nnz = h->nnz;
  return MPG_SUCCESS;
}

int mpg_handle_store_ms(mpg_handle h, float *ms_total) {
  MPG_ARG(h, "mpg_handle_store_ms: NULL handle");
  if (ms_total) *ms_total = h->store_ms;
  return MPG_SUCCESS;
}

int mpg_handle_store_path(mpg_handle h, int *candidates) {
  MPG_ARG(h, "mpg_handle_store_path: NULL handle");
  if (candidates) *candidates = h->store_path;
  return MPG_SUCCESS;
}

int mpg_handle_store_stats(mpg_handle h, int64_t *stats_host, int n) {
  MPG_ARG(h && stats_host && n >= 1, "mpg_handle_store_stats: bad argument");
  for (int k = 0; k < n; ++k) stats_host[k] = k == 0 ? (int64_t)h->store_path : k < 8 ? h->store_stats[k] : 0;
  return MPG_SUCCESS;
}

int mpg_handle_get_weights(mpg_handle h, int32_t *idx_host, double *w_host) {
  MPG_CHECK_INIT();
  MPG_ARG(h && idx_host, "mpg_handle_get_weights: NULL argument");
  MPG_ARG(h->kind == MPG_KIND_FIXED, "mpg_handle_get_weights: handle is CSR, use mpg_handle_get_csr");
  int nz = h->nnz_per_row;
  int64_t P = h->n_dst;
  std::vector<int32_t> ti((size_t)nz * P);
  MPG_HIP(hipMemcpy(ti.data(), h->idx.p, sizeof(int32_t) * nz * P, hipMemcpyDeviceToHost));
  for (int64_t p = 0; p < P; ++p)
    for (int q = 0; q < nz; ++q) idx_host[p * nz + q] = ti[(size_t)q * P + p];
  if (w_host) {
    if (h->w.p) {
      std::vector<double> tw((size_t)nz * P);
      MPG_HIP(hipMemcpy(tw.data(), h->w.p, sizeof(double) * nz * P, hipMemcpyDeviceToHost));
      for (int64_t p = 0; p < P; ++p)
        for (int q = 0; q < nz; ++q) w_host[p * nz + q] = tw[(size_t)q * P + p];
    } else {
      for (int64_t p = 0; p < P * nz; ++p) w_host[p] = idx_host[p] >= 0 ? 1.0 : 0.0;
    }
  }
  return MPG_SUCCESS;
}

int mpg_handle_get_csr(mpg_handle h, int64_t *rowptr_host, int32_t *col_host, double *val_host) {
  MPG_CHECK_INIT();
  MPG_ARG(h && rowptr_host, "mpg_handle_get_csr: NULL argument");
  MPG_ARG(h->kind == MPG_KIND_CSR, "mpg_handle_get_csr: handle is not CSR");
  std::vector<int32_t> rp((size_t)h->n_dst + 1);
  MPG_HIP(hipMemcpy(rp.data(), h->rowptr.p, sizeof(int32_t) * (h->n_dst + 1), hipMemcpyDeviceToHost));
  for (int64_t p = 0; p <= h->n_dst; ++p) rowptr_host[p] = rp[p];
  if (col_host) MPG_HIP(hipMemcpy(col_host, h->col.p, sizeof(int32_t) * h->nnz, hipMemcpyDeviceToHost));
  if (val_host) MPG_HIP(hipMemcpy(val_host, h->val.p, sizeof(double) * h->nnz, hipMemcpyDeviceToHost));
  return MPG_SUCCESS;
}

int mpg_handle_kernel_choice(mpg_handle h, int *cell_fast_kernel, int *lev_fast_kernel, int *max_unique) {
  MPG_ARG(h, "mpg_handle_kernel_choice: NULL handle");
  if (cell_fast_kernel) *cell_fast_kernel = h->cf_choice;
  if (lev_fast_kernel) *lev_fast_kernel = h->lf_choice;
  if (max_unique) *max_unique = h->ut_max;
  return MPG_SUCCESS;
}

int mpg_handle_tile_stats(mpg_handle h, int *tile_nx, int *tile_ny, double *reuse, double *line_fill) {
  MPG_ARG(h, "mpg_handle_tile_stats: NULL handle");
  if (h->ut_rpt == 0 || h->ut_total <= 0) {
    mpg_set_error("mpg_handle_tile_stats: the handle has no tile lists (no staged Regrid has run on it yet)");
    return MPG_ERR_INVALID_ARG;
  }
  if (tile_nx) *tile_nx = (h->ut_rpt & 0xFFFFFF) / 1024;   // bit 24: alignment rule of the lists (k_apply_lfu.hip)
  if (tile_ny) *tile_ny = (h->ut_rpt & 0xFFFFFF) % 1024;
  if (reuse) *reuse = 3.0 * (double)h->n_dst / (double)h->ut_total;
  if (line_fill) *line_fill = h->ut_lines > 0 ? (double)h->ut_total / (16.0 * (double)h->ut_lines) : 0.0;
  return MPG_SUCCESS;
}

int mpg_handle_pole_count(mpg_handle h, int64_t *n_points, int *row_len) {
  MPG_ARG(h, "mpg_handle_pole_count: NULL handle");
  if (n_points) *n_points = h->n_pole;
  if (row_len) *row_len = h->pole_len;
  return MPG_SUCCESS;
}

int mpg_handle_get_pole(mpg_handle h, int32_t *dst_id_host, int32_t *src_row_start_host, double *w_pole_host) {
  MPG_CHECK_INIT();
  MPG_ARG(h, "mpg_handle_get_pole: NULL handle");
  if (h->n_pole == 0) return MPG_SUCCESS;
  if (dst_id_host) MPG_HIP(hipMemcpy(dst_id_host, h->pole_dst.p, sizeof(int32_t) * h->n_pole, hipMemcpyDeviceToHost));
  if (src_row_start_host) MPG_HIP(hipMemcpy(src_row_start_host, h->pole_src0.p, sizeof(int32_t) * h->n_pole, hipMemcpyDeviceToHost));
  if (w_pole_host) MPG_HIP(hipMemcpy(w_pole_host, h->pole_w.p, sizeof(double) * h->n_pole, hipMemcpyDeviceToHost));
  return MPG_SUCCESS;
}

// the tile lists of the LDS-staged level-fast kernel hold source ids: re-indexing a handle drops them
static void lf_invalidate(mpg_handle_s *h) {
  h->free_tile_lists();
  h->lf_choice = 0;
  h->cf_choice = 0;
}

// ---- multi-GPU halo support (kernels in k_halo.hip) ----------------------------------------------------
// The source window of a mesh (mpg_mesh_set_source_window) and the in-place re-indexing verbs below are two answers to the
// same question -- which sources does this rank hold -- and do not compose: a handle of a windowed mesh indexes relative to
// the window, which mpg_handle_rebase / _localize / mpg_halo_build would take for global ids.  They refuse such a handle;
// mpg_handle_unique_sources and mpg_handle_source_range report GLOBAL ids whatever the window is.
}  // extern "C"
int64_t mpg_handle_window_first(const mpg_handle_s *h) {   // > 0 offset of the handle's indices, 0 when its mesh is not windowed
  if (!h->cached || std::get<1>(h->key) >= 100) return 0;
  const mpg_mesh_s *m = (const mpg_mesh_s *)std::get<0>(h->key);
  const int loc = std::get<1>(h->key);
  return m->win_count[loc] >= 0 ? m->win_first[loc] : 0;
}
bool mpg_handle_is_windowed(const mpg_handle_s *h) {
  if (!h->cached || std::get<1>(h->key) >= 100) return false;
  return ((const mpg_mesh_s *)std::get<0>(h->key))->win_count[std::get<1>(h->key)] >= 0;
}
extern "C" {
#define MPG_NOT_WINDOWED(h, what)                                                                                                  \
  MPG_ARG(!mpg_handle_is_windowed(h), what ": the handle's mesh has a source window (mpg_mesh_set_source_window); a windowed mesh and " \
                                           "in-place re-indexing are alternatives -- reset the window to the whole mesh first")

int mpg_handle_unique_sources(mpg_handle h, int64_t *n_unique, int32_t *ids_host) {
  MPG_CHECK_INIT();
  MPG_ARG(h && n_unique, "mpg_handle_unique_sources: NULL argument");
  MPG_ARG(!h->localized, "mpg_handle_unique_sources: handle already localized");
  MPG_ARG(h->n_pole == 0, "mpg_handle_unique_sources: handles with pole terms (periodic Grid -> Grid) cannot be re-indexed");
  std::vector<int32_t> ids;
  int rc = mpg_k_unique_sources(h, ids, false, g_stream);
  if (rc) return rc;
  const int32_t w0 = (int32_t)mpg_handle_window_first(h);   // window-relative -> global
  if (w0)
    for (int32_t &c : ids) c += w0;
  *n_unique = (int64_t)ids.size();
  if (ids_host && !ids.empty()) memcpy(ids_host, ids.data(), sizeof(int32_t) * ids.size());
  return MPG_SUCCESS;
}

int mpg_handle_localize(mpg_handle h) {
  MPG_CHECK_INIT();
  MPG_ARG(h, "mpg_handle_localize: NULL handle");
  MPG_ARG(!h->localized, "mpg_handle_localize: handle already localized");
  MPG_ARG(h->refcount <= 1, "mpg_handle_localize: the handle is shared (a second mpg_regrid_store returned it from the cache); "
                            "re-indexing it in place would corrupt the other holder's indices -- release the other reference first");
  MPG_ARG(h->n_pole == 0, "mpg_handle_localize: handles with pole terms (periodic Grid -> Grid) cannot be re-indexed");
  MPG_NOT_WINDOWED(h, "mpg_handle_localize");
  // a localized handle no longer matches its cache key: detach it
  if (h->cached) {
    CACHE_LOCK();
    g_cache.erase(h->key);
    h->cached = false;
  }
  std::vector<int32_t> ids;
  lf_invalidate(h);
  return mpg_k_unique_sources(h, ids, true, g_stream);
}

int mpg_handle_rebase(mpg_handle h, int64_t base, int64_t n_local) {
  MPG_CHECK_INIT();
  MPG_ARG(h, "mpg_handle_rebase: NULL handle");
  MPG_ARG(!h->localized, "mpg_handle_rebase: handle already localized");
  MPG_ARG(h->refcount <= 1, "mpg_handle_rebase: the handle is shared (a second mpg_regrid_store returned it from the cache); "
                            "re-indexing it in place would corrupt the other holder's indices -- release the other reference first");
  MPG_ARG(h->n_pole == 0, "mpg_handle_rebase: handles with pole terms (periodic Grid -> Grid) cannot be re-indexed");
  MPG_ARG(base >= 0 && n_local >= 0 && n_local < 0x7fffffff, "mpg_handle_rebase: bad range");
  MPG_NOT_WINDOWED(h, "mpg_handle_rebase");
  if (h->cached) {
    CACHE_LOCK();
    g_cache.erase(h->key);
    h->cached = false;
  }
  lf_invalidate(h);
  return mpg_k_rebase(h, base, n_local, g_stream);
}

int mpg_tune(const char *key, int value) {
  MPG_ARG(key, "mpg_tune: NULL key");
  store_worker_drain();   // a Store that was begun under the old setting finishes under it: its cache key says so
  int rc = mpg_k_tune(key, value);
  if (rc) mpg_set_error("mpg_tune: unknown key or value out of range: %s=%d", key, value);
  return rc;
}

int mpg_pack_dev(const double *src_dev, int64_t n_src, int nlev, const int32_t *ids_dev, int64_t n_ids, double *dst_dev, void *hip_stream) {
  MPG_CHECK_INIT();
  MPG_ARG(src_dev && ids_dev && dst_dev, "mpg_pack_dev: NULL argument");
  return mpg_k_pack(src_dev, n_src, nlev, ids_dev, n_ids, dst_dev, (hipStream_t)hip_stream);
}

// ---- device buffers for hosts without a HIP binding of their own (the Fortran driver keeps its fields in HBM between
// the input file and the output file; ESMF owned that storage behind ESMF_FieldCreate / farrayPtr) ---------------------
int mpg_dev_alloc(int64_t nbytes, void **out_dev) {
  MPG_CHECK_INIT();
  MPG_ARG(out_dev && nbytes >= 0, "mpg_dev_alloc: bad argument");
  *out_dev = nullptr;
  if (nbytes == 0) return MPG_SUCCESS;
  if (hipMalloc(out_dev, (size_t)nbytes) != hipSuccess) {   // the library's own caches may be what fills the device
    (void)hipGetLastError();
    mpg_release_caches();
    MPG_HIP(hipMalloc(out_dev, (size_t)nbytes));
  }
  return MPG_SUCCESS;
}

int mpg_dev_free(void *dev) {
  MPG_CHECK_INIT();
  if (dev) MPG_HIP(hipFree(dev));
  return MPG_SUCCESS;
}

int mpg_dev_upload(void *dst_dev, const void *src_host, int64_t nbytes) {
  MPG_CHECK_INIT();
  MPG_ARG(nbytes >= 0 && (nbytes == 0 || (dst_dev && src_host)), "mpg_dev_upload: bad argument");
  if (nbytes) MPG_HIP(hipMemcpy(dst_dev, src_host, (size_t)nbytes, hipMemcpyHostToDevice));
  return MPG_SUCCESS;
}

int mpg_dev_download(void *dst_host, const void *src_dev, int64_t nbytes) {
  MPG_CHECK_INIT();
  MPG_ARG(nbytes >= 0 && (nbytes == 0 || (dst_host && src_dev)), "mpg_dev_download: bad argument");
  if (nbytes) MPG_HIP(hipMemcpy(dst_host, src_dev, (size_t)nbytes, hipMemcpyDeviceToHost));
  return MPG_SUCCESS;
}

// ---- output epilogues (kernels in k_post.hip) ------------------------------------------------------------
int mpg_bswap_dev(void *buf_dev, int64_t n, int elem_size, void *hip_stream) {
  MPG_CHECK_INIT();
  MPG_ARG(n >= 0 && (n == 0 || buf_dev), "mpg_bswap_dev: NULL argument");
  return mpg_k_bswap(buf_dev, n, elem_size, (hipStream_t)hip_stream);
}

int mpg_post_cast_dev(const double *src_dev, int64_t n, double scale, double offset, float *dst_dev, int dst_be, void *hip_stream) {
  MPG_CHECK_INIT();
  MPG_ARG(n >= 0 && (n == 0 || (src_dev && dst_dev)), "mpg_post_cast_dev: NULL argument");
  return mpg_k_post_cast(src_dev, n, scale, offset, dst_dev, dst_be != 0, (hipStream_t)hip_stream);
}

int mpg_post_layer_mean_dev(const double *src_dev, int nlevp1, int64_t n_pts, float *dst_dev, int dst_be, void *hip_stream) {
  MPG_CHECK_INIT();
  MPG_ARG(nlevp1 >= 2 && n_pts >= 0, "mpg_post_layer_mean_dev: needs at least two levels");
  MPG_ARG(n_pts == 0 || (src_dev && dst_dev), "mpg_post_layer_mean_dev: NULL argument");
  return mpg_k_post_layer_mean(src_dev, nlevp1, n_pts, dst_dev, dst_be != 0, (hipStream_t)hip_stream);
}

int mpg_post_ptop_dev(const double *p_hyd_dev, int nlev, int64_t n_pts, double *ptop_host, void *hip_stream) {
  MPG_CHECK_INIT();
  MPG_ARG(p_hyd_dev && ptop_host && nlev >= 1 && n_pts >= 1, "mpg_post_ptop_dev: bad argument");
  return mpg_k_post_ptop(p_hyd_dev, nlev, n_pts, ptop_host, (hipStream_t)hip_stream);
}

int mpg_post_ptop_parts_dev(const double *p_hyd_dev, int nlev, int64_t n_pts, double *vmax_host, double *candmin_host, int *has_cand_host,
                            void *hip_stream) {
  MPG_CHECK_INIT();
  MPG_ARG(p_hyd_dev && vmax_host && candmin_host && has_cand_host && nlev >= 1 && n_pts >= 1, "mpg_post_ptop_parts_dev: bad argument");
  return mpg_k_post_ptop_parts(p_hyd_dev, nlev, n_pts, vmax_host, candmin_host, has_cand_host, (hipStream_t)hip_stream);
}

// Test hook (include/mpassit_amd.h): the device-wide primitives of k_prims.hip on host arrays
int mpg_debug_scan_i32(const int32_t *in_host, int64_t n, int32_t *out_host, long long *sum_host) {
  MPG_CHECK_INIT();
  MPG_ARG(n >= 0 && (in_host || n == 0), "mpg_debug_scan_i32: bad argument");
  if (sum_host) *sum_host = 0;
  if (n == 0) return MPG_SUCCESS;
  hipStream_t s = g_stream;
  TmpBuf<int32_t> in, out;
  TmpBuf<long long> sum;
  int rc;
  if ((rc = in.alloc((size_t)n, s)) || (rc = out.alloc((size_t)n, s)) || (rc = sum.alloc(1, s))) return rc;
  MPG_HIP(hipMemcpyAsync(in.p, in_host, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, s));
  if ((rc = mpg_sum_i32_i64(in.p, n, sum.p, s)) || (rc = mpg_scan_excl_i32(in.p, out.p, n, s))) return rc;
  if (out_host) MPG_HIP(hipMemcpyAsync(out_host, out.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
  if (sum_host) MPG_HIP(hipMemcpyAsync(sum_host, sum.p, sizeof(long long), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  // in place (in == out) is what the Stores do: the same answer
  if (out_host) {
    if ((rc = mpg_scan_excl_i32(in.p, in.p, n, s))) return rc;
    std::vector<int32_t> again((size_t)n);
    MPG_HIP(hipMemcpyAsync(again.data(), in.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
    MPG_HIP(hipStreamSynchronize(s));
    if (memcmp(again.data(), out_host, sizeof(int32_t) * (size_t)n) != 0) {
      mpg_set_error("mpg_debug_scan_i32: the in-place scan differs from the out-of-place one");
      return MPG_ERR_HIP;
    }
  }
  return MPG_SUCCESS;
}

}  // extern "C"
