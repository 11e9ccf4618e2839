// How fast can the card take the STORES of a Regrid, by the shape of the tile a workgroup writes?  No loads, no arithmetic: every
// workgroup writes its tile of TW x TH points for all levels (chunks of 16, like k_apply3_lfu) into a float32 [nlev][ny][nx] result
// with non-temporal stores, lanes along x.  Against it: a plain linear fill of the same bytes.  If 64 x 8 tiles write much slower
// than the fill, the store PATTERN (256-byte pieces scattered over rows and level planes) is a ceiling of its own for the write-heavy
// shapes (global lat-lon targets: 59-94 % of the algorithmic bytes are stores), whatever the loads do.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_probe tools/store_pattern_probe.hip && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcd_remap(unsigned lin, unsigned n) {   // geom.h: one contiguous range of the work space per XCD
  unsigned q = n / 8, r = n % 8, xcd = lin % 8, k = lin / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// TW x TH points per workgroup of NT threads; thread t owns points t, t + NT, ... of the tile (x fastest); LC levels per chunk
template <int TW, int TH, int NT, int LC, bool REMAP, bool ALIGN>
__global__ __launch_bounds__(NT) void k_tiles(float *__restrict__ dst, int nx, int ny, int nlev, int ntx) {
  const unsigned tile = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int tx = tile % ntx, ty = tile / ntx;
  constexpr int RPT = TW * TH / NT;
  const int64_t P = (int64_t)nx * ny;
  for (int k0 = 0; k0 < nlev; k0 += LC) {
    const int kn = min(LC, nlev - k0);
    for (int kk = 0; kk < kn; ++kk) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const int pt = threadIdx.x + NT * r, j = ty * TH + pt / TW, i = tx * TW + pt % TW - (ALIGN ? (int)(((long long)j * nx) % 32) : 0);
        if (i >= 0 && i < nx && j < ny) __builtin_nontemporal_store((float)(k0 + kk) + 0.5f, dst + (int64_t)(k0 + kk) * P + (int64_t)j * nx + i);
      }
    }
    __syncthreads();   // the kernel's chunk barrier
  }
}

// the same with TWO adjacent points per lane (one 8-byte store: 512 bytes per wave-instruction, what a float64 result gets for free)
typedef float f2 __attribute__((ext_vector_type(2)));
template <int TW, int TH, int NT, int LC>
__global__ __launch_bounds__(NT) void k_tiles2(float *__restrict__ dst, int nx, int ny, int nlev, int ntx) {
  const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = tile % ntx, ty = tile / ntx;
  constexpr int RPT = TW * TH / NT / 2;
  const int64_t P = (int64_t)nx * ny;
  for (int k0 = 0; k0 < nlev; k0 += LC) {
    const int kn = min(LC, nlev - k0);
    for (int kk = 0; kk < kn; ++kk) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const int pt = 2 * (threadIdx.x + NT * r), j = ty * TH + pt / TW, i = tx * TW + pt % TW - (int)(((long long)j * nx) % 32 & ~1);   // (even shift: pairs stay 8-byte aligned)
        const f2 v = {(float)(k0 + kk) + 0.5f, (float)(k0 + kk) + 0.25f};
        if (i >= 0 && i + 1 < nx && j < ny) __builtin_nontemporal_store(v, (f2 *)(dst + (int64_t)(k0 + kk) * P + (int64_t)j * nx + i));
      }
    }
    __syncthreads();
  }
}

typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_fill(f4 *__restrict__ dst, int64_t n4) {
  const f4 v = {1.5f, 1.5f, 1.5f, 1.5f};
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(v, dst + i);
}

template <typename F>
static double timed(F launch, int n = 5) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int k = 0; k < n; ++k) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms / n * 1e-3;
}

template <int TW, int TH, int NT, bool REMAP, bool ALIGN = false>
static void run(const char *name, float *d, int nx, int ny, int nlev) {
  const int ntx = (nx + 31 + TW - 1) / TW, nty = (ny + TH - 1) / TH;
  const double gb = (double)nx * ny * nlev * 4 / 1e9;
  double t = timed([&] { k_tiles<TW, TH, NT, 16, REMAP, ALIGN><<<ntx * nty, NT>>>(d, nx, ny, nlev, ntx); });
  printf("  %-28s %7.0f GB/s\n", name, gb / t);
}

int main() {
  const int shapes[3][3] = {{1800, 1060, 55 * 4}, {3600, 1800, 55 * 2}, {7200, 3600, 55}};   // x fields so that every case is 1.6-5.7 GB
  for (const auto &s : shapes) {
    const int nx = s[0], ny = s[1], nlev = s[2];
    const int64_t n = (int64_t)nx * ny * nlev;
    float *d;
    CK(hipMalloc(&d, n * 4));
    printf("[%d levels][%d][%d] float32 = %.2f GB\n", nlev, ny, nx, n * 4 / 1e9);
    double t = timed([&] { k_fill<<<8192, 256>>>((f4 *)d, n / 4); });
    printf("  %-28s %7.0f GB/s\n", "linear fill (float4)", n * 4 / 1e9 / t);
    run<64, 8, 512, true>("64 x 8 tiles (the kernel's)", d, nx, ny, nlev);
    run<64, 8, 512, true, true>("64 x 8, rows shifted to 128 B", d, nx, ny, nlev);
    run<64, 8, 512, false>("64 x 8, no XCD remap", d, nx, ny, nlev);
    run<64, 4, 256, true>("64 x 4 tiles, 256 threads", d, nx, ny, nlev);
    run<64, 16, 512, true>("64 x 16 tiles", d, nx, ny, nlev);
    run<128, 4, 512, true>("128 x 4 tiles", d, nx, ny, nlev);
    run<256, 2, 512, true>("256 x 2 tiles", d, nx, ny, nlev);
    run<512, 1, 512, true>("512 x 1 tiles", d, nx, ny, nlev);
    {
      const double gb = (double)nx * ny * nlev * 4 / 1e9;
      int ntx = (nx + 31 + 127) / 128, nty = (ny + 7) / 8;
      double t2 = timed([&] { k_tiles2<128, 8, 512, 16><<<ntx * nty, 512>>>(d, nx, ny, nlev, ntx); });
      printf("  %-28s %7.0f GB/s\n", "128 x 8, 2 per lane, shifted", gb / t2);
      ntx = (nx + 31 + 127) / 128, nty = (ny + 3) / 4;
      t2 = timed([&] { k_tiles2<128, 4, 256, 16><<<ntx * nty, 256>>>(d, nx, ny, nlev, ntx); });
      printf("  %-28s %7.0f GB/s\n", "128 x 4, 2/lane, 256, shifted", gb / t2);
    }
    CK(hipFree(d));
  }
  return 0;
}
