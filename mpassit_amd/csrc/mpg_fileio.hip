// mpg_file_to_dev / mpg_dev_to_file: a byte range of a file <-> device memory, with nothing converted on the way.
//
// NetCDF classic variables are contiguous big-endian byte ranges (ncio_var_extent).  The reference reads them into host
// arrays with nf90_get_var (converting and byte-swapping every value on one core, input_data.F90:630) and writes the
// results back through nf90_put_var (write_data.F90:1008-1475).  Here the bytes travel file -> pinned staging -> HBM (and
// back) untouched and are turned around on the GPU (mpg_bswap_dev); the Regrid widens / narrows them in its loads and
// stores.  The host side of such a transfer is page-cache work: pread / pwrite are kernel copies that one core cannot
// do at PCIe speed, so a range is cut into chunks that a few threads move independently, each with two pinned staging
// buffers and its own stream (DMA of one chunk overlaps the page-cache copy of the next).
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <thread>

#include "mpg_internal.h"

namespace {
constexpr int NTHREAD = 8;   // most a direction uses; how many it does use: Lane::nt
constexpr size_t CHUNK = (size_t)32 << 20;

// Staging of one direction: NTHREAD x 2 pinned buffers and NTHREAD streams, created on first use and kept for the
// process.  One transfer per direction at a time (a reader and a writer may run concurrently, e.g. a prefetching reader
// thread beside the thread that writes the previous time level).
struct Lane {
  std::mutex mu;
  bool ready = false;
  size_t bufsize = CHUNK;   // bytes of each pinned staging buffer
  int nt = 4;   // worker threads of this direction (read: 8, the page-cache copy of pread scales; write: 4, pwrite into one tmpfs file does not)
  void *buf[NTHREAD][2] = {};
  hipStream_t stream[NTHREAD] = {};
  hipEvent_t done[NTHREAD][2] = {};
  int init_all() {
    for (int t = 0; t < nt; ++t) {   // only the lanes this direction uses: 2 x 32 MB of pinned memory each
      MPG_HIP(hipStreamCreateWithFlags(&stream[t], hipStreamNonBlocking));
      for (int b = 0; b < 2; ++b) {
        MPG_HIP(hipHostMalloc(&buf[t][b], bufsize, hipHostMallocDefault));
        MPG_HIP(hipEventCreateWithFlags(&done[t][b], hipEventDisableTiming));
      }
    }
    return MPG_SUCCESS;
  }
  int init() {   // caller holds mu
    if (ready) return MPG_SUCCESS;
    int rc = init_all();
    if (rc) {
      free_all();   // a partial set (e.g. pinned memory exhausted half way) is given back, the next call starts clean
      return rc;
    }
    ready = true;
    return MPG_SUCCESS;
  }
  void release() {
    std::lock_guard<std::mutex> lock(mu);
    free_all();
  }
  void free_all() {
    for (int t = 0; t < NTHREAD; ++t) {
      if (stream[t]) (void)hipStreamDestroy(stream[t]);
      stream[t] = nullptr;
      for (int b = 0; b < 2; ++b) {
        if (buf[t][b]) (void)hipHostFree(buf[t][b]);
        if (done[t][b]) (void)hipEventDestroy(done[t][b]);
        buf[t][b] = nullptr;
        done[t][b] = nullptr;
      }
    }
    ready = false;
  }
};
Lane g_read, g_write;
struct LaneSetup { LaneSetup() { g_read.nt = 8; const char *e = getenv("MPG_IO_READ_THREADS"); if (e && atoi(e) >= 1 && atoi(e) <= NTHREAD) g_read.nt = atoi(e); } } g_lane_setup;

int xfer_full(int fd, bool writing, char *p, size_t n, off_t off) {
  while (n > 0) {
    ssize_t k = writing ? pwrite(fd, p, n, off) : pread(fd, p, n, off);
    if (k <= 0) return -1;
    p += k;
    off += k;
    n -= (size_t)k;
  }
  return 0;
}

// error codes of a worker: 0 ok, 1 file I/O, 2 HIP
int transfer(Lane &L, int device, int fd, bool to_dev, int64_t offset, int64_t nbytes, char *dev) {
  const size_t chunk = CHUNK;
  const int64_t nchunk = (nbytes + (int64_t)chunk - 1) / (int64_t)chunk;
  std::atomic<int64_t> next(0);
  std::atomic<int> err(0);
  auto work = [&](int t) {
    if (hipSetDevice(device) != hipSuccess) {
      err = 2;
      return;
    }
    bool used[2] = {false, false};
    for (int turn = 0;; ++turn) {
      const int64_t c = next.fetch_add(1);
      if (c >= nchunk || err.load()) break;
      const int b = turn & 1;
      const size_t lo = (size_t)c * chunk, n = (size_t)(nbytes - (int64_t)lo < (int64_t)chunk ? nbytes - (int64_t)lo : (int64_t)chunk);
      if (used[b] && hipEventSynchronize(L.done[t][b]) != hipSuccess) { err = 2; break; }  // the buffer's previous DMA
      if (to_dev) {
        if (xfer_full(fd, false, (char *)L.buf[t][b], n, (off_t)(offset + (int64_t)lo))) { err = 1; break; }
        if (hipMemcpyAsync(dev + lo, L.buf[t][b], n, hipMemcpyHostToDevice, L.stream[t]) != hipSuccess ||
            hipEventRecord(L.done[t][b], L.stream[t]) != hipSuccess) { err = 2; break; }
        used[b] = true;
      } else {
        if (hipMemcpyAsync(L.buf[t][b], dev + lo, n, hipMemcpyDeviceToHost, L.stream[t]) != hipSuccess ||
            hipStreamSynchronize(L.stream[t]) != hipSuccess) { err = 2; break; }
        if (xfer_full(fd, true, (char *)L.buf[t][b], n, (off_t)(offset + (int64_t)lo))) { err = 1; break; }
      }
    }
    if (hipStreamSynchronize(L.stream[t]) != hipSuccess) err = 2;
  };
  const int nt = nchunk < L.nt ? (int)nchunk : L.nt;
  std::thread th[NTHREAD];
  for (int t = 1; t < nt; ++t) th[t] = std::thread(work, t);
  work(0);
  for (int t = 1; t < nt; ++t) th[t].join();
  return err.load();
}

int run(bool to_dev, const char *path, int64_t offset, int64_t nbytes, void *dev, hipStream_t s) {
  if (nbytes == 0) return MPG_SUCCESS;
  Lane &L = to_dev ? g_read : g_write;
  std::lock_guard<std::mutex> lock(L.mu);
  int rc = L.init();
  if (rc) return rc;
  MPG_HIP(hipStreamSynchronize(s));  // to_dev: earlier readers of the buffer are done; to file: its producer is
  const int fd = open(path, to_dev ? O_RDONLY : O_WRONLY);
  if (fd < 0) {
    mpg_set_error("%s: cannot open %s", to_dev ? "mpg_file_to_dev" : "mpg_dev_to_file", path);
    return MPG_ERR_INVALID_ARG;
  }
  int device = 0;
  (void)hipGetDevice(&device);
  const int e = transfer(L, device, fd, to_dev, offset, nbytes, (char *)dev);
  close(fd);
  if (e == 1) {
    mpg_set_error("%s: short %s on %s (offset %lld, %lld bytes)", to_dev ? "mpg_file_to_dev" : "mpg_dev_to_file", to_dev ? "read" : "write", path,
                  (long long)offset, (long long)nbytes);
    return MPG_ERR_INVALID_ARG;
  }
  if (e == 2) {
    mpg_set_error("%s: HIP transfer failed", to_dev ? "mpg_file_to_dev" : "mpg_dev_to_file");
    return MPG_ERR_HIP;
  }
  return MPG_SUCCESS;
}
}  // namespace

// mpg_finalize: the staging buffers and streams belong to the device that was current when they were made
void mpg_fileio_release() {
  g_read.release();
  g_write.release();
}

extern "C" int mpg_file_to_dev(const char *path, int64_t offset, int64_t nbytes, void *dst_dev, void *hip_stream) {
  MPG_CHECK_INIT();
  MPG_ARG(path && offset >= 0 && nbytes >= 0 && (dst_dev || nbytes == 0), "mpg_file_to_dev: bad argument");
  return run(true, path, offset, nbytes, dst_dev, (hipStream_t)hip_stream);
}

extern "C" int mpg_dev_to_file(const char *path, int64_t offset, int64_t nbytes, const void *src_dev, void *hip_stream) {
  MPG_CHECK_INIT();
  MPG_ARG(path && offset >= 0 && nbytes >= 0 && (src_dev || nbytes == 0), "mpg_dev_to_file: bad argument");
  return run(false, path, offset, nbytes, (void *)src_dev, (hipStream_t)hip_stream);
}
