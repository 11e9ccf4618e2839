// What the PCIe link gives a host-array caller on this box (numbers behind mpg_hostpipe.hip's design):
// pageable vs page-locked buffers, one direction vs both at once, one vs two copying threads per direction, and the cost
// of hipMalloc / hipHostRegister for buffers of the sizes one C4 field needs (1.32 GB up, 0.84 GB down).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pcie_probe tools/pcie_probe.hip -lpthread && /tmp/pcie_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
  const size_t NU = (size_t)3001656 * 55 * 8, ND = (size_t)1908000 * 55 * 8;
  char *hu = (char *)malloc(NU), *hd = (char *)malloc(ND);
  memset(hu, 1, NU);
  memset(hd, 2, ND);
  char *du, *dd;
  double t0 = now();
  CK(hipMalloc(&du, NU));
  CK(hipMalloc(&dd, ND));
  printf("hipMalloc 1.32 GB + 0.84 GB: %.2f ms\n", (now() - t0) * 1e3);
  CK(hipMemset(du, 0, NU));
  CK(hipMemset(dd, 0, ND));
  CK(hipDeviceSynchronize());
  auto up = [&](size_t off, size_t n) { CK(hipMemcpy(du + off, hu + off, n, hipMemcpyHostToDevice)); };
  auto down = [&](size_t off, size_t n) { CK(hipMemcpy(hd + off, dd + off, n, hipMemcpyDeviceToHost)); };
  for (int pinned = 0; pinned < 2; ++pinned) {
    if (pinned) {
      t0 = now();
      CK(hipHostRegister(hu, NU, hipHostRegisterDefault));
      CK(hipHostRegister(hd, ND, hipHostRegisterDefault));
      printf("hipHostRegister of both buffers: %.1f ms\n", (now() - t0) * 1e3);
    }
    const char *tag = pinned ? "page-locked" : "pageable   ";
    for (int rep = 0; rep < 2; ++rep) {
      t0 = now(); up(0, NU); double tu = now() - t0;
      t0 = now(); down(0, ND); double td = now() - t0;
      t0 = now();
      { std::thread a([&] { CK(hipSetDevice(0)); up(0, NU); }), b([&] { CK(hipSetDevice(0)); down(0, ND); }); a.join(); b.join(); }
      double tb = now() - t0;
      t0 = now();
      { std::thread a([&] { CK(hipSetDevice(0)); up(0, NU / 2); }), a2([&] { CK(hipSetDevice(0)); up(NU / 2, NU - NU / 2); }),
            b([&] { CK(hipSetDevice(0)); down(0, ND / 2); }), b2([&] { CK(hipSetDevice(0)); down(ND / 2, ND - ND / 2); });
        a.join(); a2.join(); b.join(); b2.join(); }
      double tb2 = now() - t0;
      printf("%s: up %.1f ms (%.1f GB/s)  down %.1f ms (%.1f GB/s)  both at once %.1f ms (%.1f GB/s total, %.0f fields/s)  "
             "both, two threads each %.1f ms (%.0f fields/s)\n", tag, tu * 1e3, NU / tu / 1e9, td * 1e3, ND / td / 1e9, tb * 1e3,
             (NU + ND) / tb / 1e9, 1.0 / tb, tb2 * 1e3, 1.0 / tb2);
    }
  }
  // both directions at once with ASYNCHRONOUS copies on two streams (separate DMA engines), page-locked and pageable
  {
    hipStream_t su, sd;
    CK(hipStreamCreateWithFlags(&su, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep) {
      t0 = now();
      CK(hipMemcpyAsync(du, hu, NU, hipMemcpyHostToDevice, su));
      CK(hipMemcpyAsync(hd, dd, ND, hipMemcpyDeviceToHost, sd));
      CK(hipStreamSynchronize(su));
      CK(hipStreamSynchronize(sd));
      double t = now() - t0;
      printf("page-locked, async on two streams: both at once %.1f ms (%.1f GB/s total, %.0f fields/s)\n", t * 1e3, (NU + ND) / t / 1e9, 1.0 / t);
    }
    CK(hipHostUnregister(hu));
    CK(hipHostUnregister(hd));
    for (int rep = 0; rep < 2; ++rep) {
      t0 = now();
      { std::thread a([&] { CK(hipSetDevice(0)); CK(hipMemcpyAsync(du, hu, NU, hipMemcpyHostToDevice, su)); CK(hipStreamSynchronize(su)); }),
            b([&] { CK(hipSetDevice(0)); CK(hipMemcpyAsync(hd, dd, ND, hipMemcpyDeviceToHost, sd)); CK(hipStreamSynchronize(sd)); });
        a.join(); b.join(); }
      double t = now() - t0;
      printf("pageable, async on two streams from two threads: both at once %.1f ms (%.1f GB/s total, %.0f fields/s)\n", t * 1e3, (NU + ND) / t / 1e9, 1.0 / t);
    }
    // registering in 64 MB pieces while copying (pipelined page-locking): is it cheaper than one big hipHostRegister?
    t0 = now();
    const size_t PIECE = (size_t)64 << 20;
    for (size_t off = 0; off < NU; off += PIECE) CK(hipHostRegister(hu + off, NU - off < PIECE ? NU - off : PIECE, hipHostRegisterDefault));
    printf("hipHostRegister of the 1.32 GB buffer in 64 MB pieces: %.1f ms\n", (now() - t0) * 1e3);
    for (size_t off = 0; off < NU; off += PIECE) CK(hipHostUnregister(hu + off));
    CK(hipHostRegister(hu, NU, hipHostRegisterDefault));
    CK(hipHostRegister(hd, ND, hipHostRegisterDefault));
  }
  // chunked pinned staging: what a library-owned pair of page-locked bounce buffers + host memcpy threads would give
  {
    const size_t CH = (size_t)64 << 20;
    char *st[2];
    CK(hipHostMalloc(&st[0], CH, hipHostMallocDefault));
    CK(hipHostMalloc(&st[1], CH, hipHostMallocDefault));
    CK(hipHostUnregister(hu));
    CK(hipHostUnregister(hd));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    t0 = now();
    int q = 0;
    for (size_t off = 0; off < NU; off += CH, q ^= 1) {
      size_t n = NU - off < CH ? NU - off : CH;
      memcpy(st[q], hu + off, n);
      CK(hipMemcpyAsync(du + off, st[q], n, hipMemcpyHostToDevice, s));
      if (off >= CH) { /* the other buffer is free once its copy is done */ }
      CK(hipStreamSynchronize(s));
    }
    double t = now() - t0;
    printf("staged through two 64 MB page-locked buffers, one memcpy thread, serialised: up %.1f ms (%.1f GB/s)\n", t * 1e3, NU / t / 1e9);
    t0 = now();
    memcpy(st[0], hu, CH);
    printf("host memcpy of 64 MB: %.2f ms (%.1f GB/s)\n", (now() - t0) * 1e3, CH / (now() - t0) / 1e9);
  }
  return 0;
}
