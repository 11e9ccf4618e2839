// mpg_regrid_typed: the host-pointer Regrid with fused ingest / egress, pipelined over PCIe.
//
// The reference's data path per field is file -> host array -> (ESMF) -> host array -> file.  A host that keeps that
// shape (the Fortran driver, io_nc.py) is bound by the PCIe link, not by the kernel (C4: 1.32 GB up + 0.84 GB down per
// float64 field at ~55 GB/s = 39 ms, against 0.37 ms of kernel time).  Two levers, both here:
//   * element types: float32 sources as the MPAS file stores them and float32 results as the output file stores them
//     cross the link (half the bytes); the arithmetic stays float64 on the device (mpg_regrid_typed_dev semantics,
//     including the writer's scale / offset epilogue);
//   * full duplex: the field is cut into chunks of levels (cell-fast) or fields; the calling thread uploads chunk c+1
//     and launches its kernel while a helper thread downloads chunk c-1, three device slots in flight, so upload and
//     download overlap instead of alternating.
// Caller buffers are ordinary pageable memory (Fortran allocatables, numpy arrays).  What the link gives such a caller on
// MI355X (tools/pcie_probe.hip, profiles/r02_pcie_probe.txt): 57 GB/s in one direction, and 90-93 GB/s in BOTH at once only
// when the two directions are asynchronous copies on two separate non-blocking streams issued from two threads (blocking
// hipMemcpy calls share the null stream and serialise: 57 GB/s in total, however many threads); page-locking the caller's
// buffers first costs as much as the transfer itself (hipHostRegister: ~50 ms per GB), so they are left alone -- the
// runtime pins a pageable range on first use and remembers it, repeated calls on the same buffers run at the pinned rate.
// hipMalloc is expensive too (55-145 ms for the 2.2 GB of one float64 field), so the device slots, streams and the
// download thread's stream are created once per process and reused by every call (released by mpg_finalize).
#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>

#include "mpg_internal.h"

namespace {
struct Chunk {           // one upload + Regrid + download of the pipeline: nlev levels of ONE field
  const char *src;       // host
  size_t src_n;          // elements
  char *dst;             // host
  size_t dst_n;
  int nlev, nfields;     // nfields is 1
  double offset;         // epilogue offset of the field the chunk belongs to
};
constexpr int NSLOT = 3;

// per-process transfer resources, grown on demand, reused by every mpg_regrid / mpg_regrid_typed call
struct Pipe {
  std::mutex mu;                       // one host-path Regrid at a time
  hipStream_t s_up = nullptr, s_k = nullptr, s_down = nullptr;
  char *dsrc[NSLOT] = {}, *ddst[NSLOT] = {};
  size_t cap_s = 0, cap_d = 0;
  int ensure(size_t need_s, size_t need_d) {
    if (!s_up) MPG_HIP(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking));
    if (!s_k) MPG_HIP(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking));
    if (!s_down) MPG_HIP(hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking));
    if (need_s > cap_s) {
      for (int q = 0; q < NSLOT; ++q) {
        if (dsrc[q]) (void)hipFree(dsrc[q]);
        dsrc[q] = nullptr;
      }
      cap_s = 0;
      for (int q = 0; q < NSLOT; ++q) MPG_HIP(hipMalloc((void **)&dsrc[q], need_s));
      cap_s = need_s;
    }
    if (need_d > cap_d) {
      for (int q = 0; q < NSLOT; ++q) {
        if (ddst[q]) (void)hipFree(ddst[q]);
        ddst[q] = nullptr;
      }
      cap_d = 0;
      for (int q = 0; q < NSLOT; ++q) MPG_HIP(hipMalloc((void **)&ddst[q], need_d));
      cap_d = need_d;
    }
    return MPG_SUCCESS;
  }
  void release() {
    std::lock_guard<std::mutex> lock(mu);
    for (int q = 0; q < NSLOT; ++q) {
      if (dsrc[q]) (void)hipFree(dsrc[q]);
      if (ddst[q]) (void)hipFree(ddst[q]);
      dsrc[q] = ddst[q] = nullptr;
    }
    cap_s = cap_d = 0;
    if (s_up) (void)hipStreamDestroy(s_up);
    if (s_k) (void)hipStreamDestroy(s_k);
    if (s_down) (void)hipStreamDestroy(s_down);
    s_up = s_k = s_down = nullptr;
  }
};
Pipe g_pipe;
}  // namespace

int mpg_device_index();  // mpg_api.hip
void mpg_hostpipe_release() { g_pipe.release(); }

// fields[f] -> (source, destination, offset): the chunk plan (~96 MB of source per chunk; a file-order field is one chunk)
static void plan_field(std::vector<Chunk> &plan, const mpg_handle_s *h, const void *src, int src_layout, int nlev, void *dst, size_t es, size_t ed,
                       double offset) {
  const size_t ns = (size_t)h->n_src, P = (size_t)h->n_dst;
  if (src_layout == MPG_LAYOUT_CELL_FAST) {
    int lch = ns ? (int)((96u << 20) / (ns * es)) : nlev;
    lch = lch < 1 ? 1 : (lch > nlev ? nlev : lch);
    for (int l0 = 0; l0 < nlev; l0 += lch) {
      const int l1 = l0 + lch < nlev ? l0 + lch : nlev;
      plan.push_back({(const char *)src + (size_t)l0 * ns * es, (size_t)(l1 - l0) * ns, (char *)dst + (size_t)l0 * P * ed, (size_t)(l1 - l0) * P, l1 - l0, 1, offset});
    }
  } else {
    plan.push_back({(const char *)src, (size_t)nlev * ns, (char *)dst, (size_t)nlev * P, nlev, 1, offset});
  }
}
static int run_plan(mpg_handle h, const std::vector<Chunk> &plan, int src_f32, int src_layout, int dst_f32, double scale);

extern "C" int mpg_regrid_typed(mpg_handle h, const void *src_host, int src_f32, int src_layout, int nlev, int nfields, void *dst_host,
                                int dst_f32, double scale, double offset) {
  MPG_CHECK_INIT();
  MPG_ARG(h && dst_host && (src_host || h->n_src == 0), "mpg_regrid_typed: NULL argument");
  MPG_ARG(nlev >= 1 && nfields >= 1, "mpg_regrid_typed: nlev and nfields must be >= 1");
  MPG_ARG(src_layout == MPG_LAYOUT_CELL_FAST || src_layout == MPG_LAYOUT_LEV_FAST, "mpg_regrid_typed: bad src_layout");
  MPG_ARG(src_f32 >= 0 && src_f32 <= 3 && dst_f32 >= 0 && dst_f32 <= 3, "mpg_regrid_typed: src_type / dst_type must be MPG_TYPE_F64 or MPG_TYPE_F32, optionally | MPG_TYPE_BE");
  const size_t es = (src_f32 & MPG_TYPE_F32) ? 4 : 8, ed = (dst_f32 & MPG_TYPE_F32) ? 4 : 8;
  const size_t ns = (size_t)h->n_src, P = (size_t)h->n_dst;
  std::vector<Chunk> plan;
  for (int f = 0; f < nfields; ++f)
    plan_field(plan, h, (const char *)src_host + (size_t)f * nlev * ns * es, src_layout, nlev, (char *)dst_host + (size_t)f * nlev * P * ed, es, ed, offset);
  return run_plan(h, plan, src_f32, src_layout, dst_f32, scale);
}

// ESMF_FieldBundleRegrid for a host that holds its fields as separate arrays (the reference's own shape: one ESMF field per
// variable, interp.F90:240-254): ALL fields of the bundle through one pipeline, so that the upload of field k + 1, the Regrid of
// field k and the download of field k - 1 overlap -- a file-order field handed over alone is one chunk and runs upload, kernel,
// download one after the other (configuration 4, float64: 39 ms per field against 23 in a bundle).
extern "C" int mpg_regrid_bundle_typed(mpg_handle h, int nfields, const void *const *src_host, int src_f32, int src_layout, int nlev,
                                       void *const *dst_host, int dst_f32, double scale, const double *offsets) {
  MPG_CHECK_INIT();
  MPG_ARG(h && src_host && dst_host, "mpg_regrid_bundle_typed: NULL argument");
  MPG_ARG(nlev >= 1 && nfields >= 1, "mpg_regrid_bundle_typed: nlev and nfields must be >= 1");
  MPG_ARG(src_layout == MPG_LAYOUT_CELL_FAST || src_layout == MPG_LAYOUT_LEV_FAST, "mpg_regrid_bundle_typed: bad src_layout");
  MPG_ARG(src_f32 >= 0 && src_f32 <= 3 && dst_f32 >= 0 && dst_f32 <= 3, "mpg_regrid_bundle_typed: src_type / dst_type must be MPG_TYPE_F64 or MPG_TYPE_F32, optionally | MPG_TYPE_BE");
  for (int f = 0; f < nfields; ++f) MPG_ARG(dst_host[f] && (src_host[f] || h->n_src == 0), "mpg_regrid_bundle_typed: NULL field pointer");
  const size_t es = (src_f32 & MPG_TYPE_F32) ? 4 : 8, ed = (dst_f32 & MPG_TYPE_F32) ? 4 : 8;
  std::vector<Chunk> plan;
  for (int f = 0; f < nfields; ++f) plan_field(plan, h, src_host[f], src_layout, nlev, dst_host[f], es, ed, offsets ? offsets[f] : 0.0);
  return run_plan(h, plan, src_f32, src_layout, dst_f32, scale);
}

static int run_plan(mpg_handle h, const std::vector<Chunk> &plan, int src_f32, int src_layout, int dst_f32, double scale) {
  const size_t es = (src_f32 & MPG_TYPE_F32) ? 4 : 8, ed = (dst_f32 & MPG_TYPE_F32) ? 4 : 8;
  const size_t ns = (size_t)h->n_src;
  size_t max_s = 0, max_d = 0;
  for (const Chunk &c : plan) {
    max_s = c.src_n > max_s ? c.src_n : max_s;
    max_d = c.dst_n > max_d ? c.dst_n : max_d;
  }
  const int nslot = NSLOT;
  std::lock_guard<std::mutex> pipe_lock(g_pipe.mu);
  int rc = g_pipe.ensure(max_s * es + 16, max_d * ed + 16);
  if (rc) return rc;
  char *const *dsrc = g_pipe.dsrc, *const *ddst = g_pipe.ddst;
  struct Res {  // events of one call, released on every exit path
    std::vector<hipEvent_t> up, done;
    ~Res() {
      for (hipEvent_t e : up) if (e) (void)hipEventDestroy(e);
      for (hipEvent_t e : done) if (e) (void)hipEventDestroy(e);
    }
  } res;
  res.up.assign(plan.size(), nullptr);
  res.done.assign(plan.size(), nullptr);
  for (size_t c = 0; c < plan.size(); ++c) {
    MPG_HIP(hipEventCreateWithFlags(&res.up[c], hipEventDisableTiming));
    MPG_HIP(hipEventCreateWithFlags(&res.done[c], hipEventDisableTiming));
  }
  hipStream_t s_up = g_pipe.s_up, s_k = g_pipe.s_k, s_down = g_pipe.s_down;
  std::vector<hipEvent_t> &up = res.up, &done = res.done;
  // Only the cells the handle references cross the link: [a, b) = its source range (a regional grid under a global mesh reads a
  // few per cent of the cells, a regional mesh that is larger than its grid loses its rim).  The device slab keeps its full
  // pitch -- the kernels index cells as ever -- and what lies outside [a, b) is never read.  (The pole caps of a periodic
  // Grid -> Grid handle read whole rows beside the indexed points: such a handle uploads everything.)
  int64_t ra = 0, rb = (int64_t)ns;
  if (ns > 0 && h->n_pole == 0) {
    if ((rc = mpg_k_source_range(h, &ra, &rb, s_k))) return rc;
    if (rb <= ra) ra = rb = 0;                                   // nothing mapped: nothing to upload
    if ((double)(rb - ra) > 0.97 * (double)ns) { ra = 0; rb = (int64_t)ns; }   // not worth the extra copies
  }
  const bool trimmed = ra != 0 || rb != (int64_t)ns;
  std::atomic<int> produced{0}, consumed{0}, err{0};
  const int dev = mpg_device_index();
  std::thread down([&]() {
    if (hipSetDevice(dev) != hipSuccess) { err = MPG_ERR_HIP; return; }
    for (size_t c = 0; c < plan.size(); ++c) {
      while (produced.load(std::memory_order_acquire) <= (int)c) {
        if (err.load()) return;
        std::this_thread::yield();
      }
      // asynchronous copy on the download stream of its own + wait: runs beside the uploads of s_up (two DMA directions at once)
      if (hipStreamWaitEvent(s_down, done[c], 0) != hipSuccess ||
          hipMemcpyAsync(plan[c].dst, ddst[c % nslot], plan[c].dst_n * ed, hipMemcpyDeviceToHost, s_down) != hipSuccess ||
          hipStreamSynchronize(s_down) != hipSuccess) {
        err = MPG_ERR_HIP;
        return;
      }
      consumed.store((int)c + 1, std::memory_order_release);
    }
  });
  rc = MPG_SUCCESS;
  for (size_t c = 0; c < plan.size() && !rc && !err.load(); ++c) {
    const int q = (int)(c % nslot);
    while ((int)c >= nslot && consumed.load(std::memory_order_acquire) <= (int)c - nslot && !err.load()) std::this_thread::yield();
    if (plan[c].src_n && !trimmed) {
      if (hipMemcpyAsync(dsrc[q], plan[c].src, plan[c].src_n * es, hipMemcpyHostToDevice, s_up) != hipSuccess) rc = MPG_ERR_HIP;
    } else if (plan[c].src_n && rb > ra) {
      const char *hsrc = plan[c].src;
      if (src_layout == MPG_LAYOUT_LEV_FAST) {                   // [cell][lev]: the referenced rows are one block
        const size_t o = (size_t)ra * (size_t)plan[c].nlev * es, nb = (size_t)(rb - ra) * (size_t)plan[c].nlev * es;
        for (int f = 0; f < plan[c].nfields && !rc; ++f)
          if (hipMemcpyAsync(dsrc[q] + (size_t)f * plan[c].nlev * ns * es + o, hsrc + (size_t)f * plan[c].nlev * ns * es + o, nb, hipMemcpyHostToDevice,
                             s_up) != hipSuccess)
            rc = MPG_ERR_HIP;
      } else {                                                   // [lev][cell]: one run per level
        const size_t o = (size_t)ra * es, nb = (size_t)(rb - ra) * es;
        for (int l = 0; l < plan[c].nlev * plan[c].nfields && !rc; ++l)
          if (hipMemcpyAsync(dsrc[q] + (size_t)l * ns * es + o, hsrc + (size_t)l * ns * es + o, nb, hipMemcpyHostToDevice, s_up) != hipSuccess)
            rc = MPG_ERR_HIP;
      }
    }
    if (!rc && (hipEventRecord(up[c], s_up) != hipSuccess || hipStreamWaitEvent(s_k, up[c], 0) != hipSuccess)) rc = MPG_ERR_HIP;
    if (!rc) rc = mpg_k_apply_typed(h, dsrc[q], src_f32, src_layout, plan[c].nlev, plan[c].nfields, ddst[q], dst_f32, scale, plan[c].offset, s_k);
    if (!rc && hipEventRecord(done[c], s_k) != hipSuccess) rc = MPG_ERR_HIP;
    if (!rc) produced.store((int)c + 1, std::memory_order_release);
  }
  if (rc) err = rc;
  down.join();
  if (!rc && err.load()) rc = err.load();
  (void)hipStreamSynchronize(s_k);
  (void)hipStreamSynchronize(s_up);
  (void)hipStreamSynchronize(s_down);
  if (rc == MPG_ERR_HIP) mpg_set_error("mpg_regrid_typed: a HIP call of the transfer pipeline failed: %s", hipGetErrorString(hipGetLastError()));
  return rc;
}

// ---- the wind chain for a host that holds its fields in HOST arrays (interp.F90:291-328 as the reference runs it: farrayPtr
// in, farrayPtr out).  The three separate calls (mpg_rotate_winds + two mpg_regrid) move the mass winds over the link twice in
// each direction -- up and down for the in-place rotation, up again for each destaggering -- and the staggered winds down once:
// 4 fields up, 4 down.  Here the earth-relative mass winds go up ONCE in chunks of levels, k_wind_destagger runs on each chunk,
// and only U and V (plus, on request, the rotated mass winds) come down: 2 up, 2 down, both directions busy at once (the pipeline
// of run_plan above: upload of chunk c + 1 and its kernel on the calling thread, download of chunk c - 1 on a helper thread).
extern "C" int mpg_wind_destagger(mpg_handle h1, mpg_handle h2, const double *cosa_host, const double *sina_host, const double *umass_host,
                                  const double *vmass_host, int nlev, void *u_host, void *v_host, int dst_type, double *umass_rot_host,
                                  double *vmass_rot_host) {
  MPG_CHECK_INIT();
  MPG_ARG(h1 || h2, "mpg_wind_destagger: no handle");
  MPG_ARG((cosa_host == nullptr) == (sina_host == nullptr), "mpg_wind_destagger: cosa and sina come together");
  const bool rot = cosa_host != nullptr;
  MPG_ARG(!rot || (h1 && h2), "mpg_wind_destagger: the rotation needs both components (interp.F90:291)");
  MPG_ARG((!h1 || u_host) && (!h2 || v_host), "mpg_wind_destagger: NULL destination");
  MPG_ARG((!(h1 || rot) || umass_host) && (!(h2 || rot) || vmass_host), "mpg_wind_destagger: NULL mass field");
  MPG_ARG(nlev >= 1, "mpg_wind_destagger: nlev must be >= 1");
  MPG_ARG(dst_type >= 0 && dst_type <= 3, "mpg_wind_destagger: dst_type must be MPG_TYPE_F64 or MPG_TYPE_F32, optionally | MPG_TYPE_BE");
  MPG_ARG(rot || (!umass_rot_host && !vmass_rot_host), "mpg_wind_destagger: rotated mass winds asked for without a rotation");
  const int nx = h2 ? h2->nx_dst : h1->nx_dst - 1, ny = h2 ? h2->ny_dst - 1 : h1->ny_dst;
  MPG_ARG(nx >= 1 && ny >= 1, "mpg_wind_destagger: the handles are not the CENTER -> EDGE1 / EDGE2 pair of one grid");
  const size_t NP = (size_t)nx * ny, P1 = (size_t)(nx + 1) * ny, P2 = (size_t)nx * (ny + 1), ed = (dst_type & MPG_TYPE_F32) ? 4 : 8;
  const bool src_u = rot || h1, src_v = rot || h2;
  int lch = (int)((96u << 20) / (2 * NP * 8));
  lch = lch < 1 ? 1 : (lch > nlev ? nlev : lch);
  const int nchunk = (nlev + lch - 1) / lch;
  // slot layout: source  um[lch][NP] | vm[lch][NP];  result  u[lch][P1] | v[lch][P2] | um_rot[lch][NP] | vm_rot[lch][NP]  (8-byte aligned parts)
  const size_t so_v = (size_t)lch * NP * 8, s_bytes = 2 * so_v;
  const size_t do_v = ((size_t)lch * P1 * ed + 7) & ~(size_t)7, do_ur = do_v + (((size_t)lch * P2 * ed + 7) & ~(size_t)7), do_vr = do_ur + so_v,
               d_bytes = do_vr + so_v;
  std::lock_guard<std::mutex> pipe_lock(g_pipe.mu);
  int rc = g_pipe.ensure(s_bytes + 16, d_bytes + 16);
  if (rc) return rc;
  hipStream_t s_up = g_pipe.s_up, s_k = g_pipe.s_k, s_down = g_pipe.s_down;
  TmpBuf<double> cs;
  if (rot) {
    if ((rc = cs.alloc(2 * NP, s_up))) return rc;
    MPG_HIP(hipMemcpyAsync(cs.p, cosa_host, 8 * NP, hipMemcpyHostToDevice, s_up));
    MPG_HIP(hipMemcpyAsync(cs.p + NP, sina_host, 8 * NP, hipMemcpyHostToDevice, s_up));
  }
  struct Res {
    std::vector<hipEvent_t> up, done;
    ~Res() {
      for (hipEvent_t e : up) if (e) (void)hipEventDestroy(e);
      for (hipEvent_t e : done) if (e) (void)hipEventDestroy(e);
    }
  } res;
  res.up.assign((size_t)nchunk, nullptr);
  res.done.assign((size_t)nchunk, nullptr);
  for (int c = 0; c < nchunk; ++c) {
    MPG_HIP(hipEventCreateWithFlags(&res.up[(size_t)c], hipEventDisableTiming));
    MPG_HIP(hipEventCreateWithFlags(&res.done[(size_t)c], hipEventDisableTiming));
  }
  std::atomic<int> produced{0}, consumed{0}, err{0};
  const int dev = mpg_device_index();
  char *const *dsrc = g_pipe.dsrc, *const *ddst = g_pipe.ddst;
  std::thread down([&]() {
    if (hipSetDevice(dev) != hipSuccess) { err = MPG_ERR_HIP; return; }
    for (int c = 0; c < nchunk; ++c) {
      while (produced.load(std::memory_order_acquire) <= c) {
        if (err.load()) return;
        std::this_thread::yield();
      }
      const int q = c % NSLOT, l0 = c * lch, nl = std::min(lch, nlev - l0);
      bool ok = hipStreamWaitEvent(s_down, res.done[(size_t)c], 0) == hipSuccess;
      if (ok && h1) ok = hipMemcpyAsync((char *)u_host + (size_t)l0 * P1 * ed, ddst[q], (size_t)nl * P1 * ed, hipMemcpyDeviceToHost, s_down) == hipSuccess;
      if (ok && h2) ok = hipMemcpyAsync((char *)v_host + (size_t)l0 * P2 * ed, ddst[q] + do_v, (size_t)nl * P2 * ed, hipMemcpyDeviceToHost, s_down) == hipSuccess;
      if (ok && umass_rot_host) ok = hipMemcpyAsync(umass_rot_host + (size_t)l0 * NP, ddst[q] + do_ur, (size_t)nl * NP * 8, hipMemcpyDeviceToHost, s_down) == hipSuccess;
      if (ok && vmass_rot_host) ok = hipMemcpyAsync(vmass_rot_host + (size_t)l0 * NP, ddst[q] + do_vr, (size_t)nl * NP * 8, hipMemcpyDeviceToHost, s_down) == hipSuccess;
      if (ok) ok = hipStreamSynchronize(s_down) == hipSuccess;
      if (!ok) { err = MPG_ERR_HIP; return; }
      consumed.store(c + 1, std::memory_order_release);
    }
  });
  rc = MPG_SUCCESS;
  for (int c = 0; c < nchunk && !rc && !err.load(); ++c) {
    const int q = c % NSLOT, l0 = c * lch, nl = std::min(lch, nlev - l0);
    while (c >= NSLOT && consumed.load(std::memory_order_acquire) <= c - NSLOT && !err.load()) std::this_thread::yield();
    if (src_u && hipMemcpyAsync(dsrc[q], umass_host + (size_t)l0 * NP, (size_t)nl * NP * 8, hipMemcpyHostToDevice, s_up) != hipSuccess) rc = MPG_ERR_HIP;
    if (!rc && src_v && hipMemcpyAsync(dsrc[q] + so_v, vmass_host + (size_t)l0 * NP, (size_t)nl * NP * 8, hipMemcpyHostToDevice, s_up) != hipSuccess) rc = MPG_ERR_HIP;
    if (!rc && (hipEventRecord(res.up[(size_t)c], s_up) != hipSuccess || hipStreamWaitEvent(s_k, res.up[(size_t)c], 0) != hipSuccess)) rc = MPG_ERR_HIP;
    if (!rc)
      rc = mpg_k_wind_destagger(h1, h2, rot ? cs.p : nullptr, rot ? cs.p + NP : nullptr, (const double *)dsrc[q], (const double *)(dsrc[q] + so_v), nl, ddst[q],
                                ddst[q] + do_v, dst_type, umass_rot_host ? (double *)(ddst[q] + do_ur) : nullptr,
                                vmass_rot_host ? (double *)(ddst[q] + do_vr) : nullptr, s_k);
    if (!rc && hipEventRecord(res.done[(size_t)c], s_k) != hipSuccess) rc = MPG_ERR_HIP;
    if (!rc) produced.store(c + 1, std::memory_order_release);
  }
  if (rc) err = rc;
  down.join();
  if (!rc && err.load()) rc = err.load();
  (void)hipStreamSynchronize(s_k);
  (void)hipStreamSynchronize(s_up);
  (void)hipStreamSynchronize(s_down);
  cs.free();
  if (rc == MPG_ERR_UNSUPPORTED) mpg_set_error("mpg_wind_destagger: the handles are not the CENTER -> EDGE1 / EDGE2 pair of one grid");
  else if (rc == MPG_ERR_HIP) mpg_set_error("mpg_wind_destagger: a HIP call of the transfer pipeline failed: %s", hipGetErrorString(hipGetLastError()));
  return rc;
}
