#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(int *p) { p[0] = 1; }
int main() {
  hipStream_t s; hipStreamCreate(&s);
  void *w; hipMalloc(&w, 1 << 20); hipFree(w);
  size_t sizes[] = {4, 1 << 20, 12 << 20, 100 << 20, 400u << 20};
  for (size_t sz : sizes) {
    double t0 = now(); void *p[8];
    for (int i = 0; i < 8; ++i) hipMalloc(&p[i], sz);
    double t1 = now();
    k<<<1, 1, 0, s>>>((int *)p[0]);
    for (int i = 0; i < 8; ++i) hipFree(p[i]);
    double t2 = now();
    printf("hipMalloc %10zu B: %8.1f us each   hipFree (one kernel pending): %8.1f us each\n", sz, (t1 - t0) / 8, (t2 - t1) / 8);
  }
  hipMemPool_t pool; hipDeviceGetDefaultMemPool(&pool, 0);
  uint64_t thr = ~0ull; hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
  for (int rep = 0; rep < 2; ++rep)
  for (size_t sz : sizes) {
    double t0 = now(); void *p[8];
    for (int i = 0; i < 8; ++i) hipMallocAsync(&p[i], sz, s);
    double t1 = now();
    k<<<1, 1, 0, s>>>((int *)p[0]);
    for (int i = 0; i < 8; ++i) hipFreeAsync(p[i], s);
    double t2 = now();
    hipStreamSynchronize(s);
    printf("rep %d hipMallocAsync %10zu B: %8.1f us each   hipFreeAsync: %8.1f us each\n", rep, sz, (t1 - t0) / 8, (t2 - t1) / 8);
  }
  int h = 0; int *d; hipMalloc(&d, 4);
  double t0 = now();
  for (int i = 0; i < 20; ++i) { k<<<1, 1, 0, s>>>(d); hipMemcpyAsync(&h, d, 4, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
  printf("kernel + 4-byte D2H + stream sync: %.1f us each\n", (now() - t0) / 20);
  return 0;
}
