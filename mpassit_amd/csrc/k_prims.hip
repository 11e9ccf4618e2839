// Device-wide primitives of the Store kernels: exclusive prefix sum of int32 counts and a 64-bit sum of int32 counts.
//
// Rounds 1-4 took these from rocPRIM.  A look inside the library's code objects (round 5, profiles/r05_init_breakdown.md) showed what
// that cost a single-shot tool (mpassit.F90:105-137): rocPRIM 4.2 instantiates every algorithm for every target architecture it
// knows and picks one at run time, so the three translation units that called exclusive_scan / reduce each carried 250-760 kernels
// nobody launches (k_store_conserve: 784 kernels, 198 KB of 293 KB of code, a 394 KB metadata note and 1.2 MB of mangled names) --
// all of it read, relocated and registered by the runtime when the code object is loaded.  The two primitives below are what the
// Stores need: counts of a few million entries, scanned once or twice per Store; both are bandwidth-trivial (26 MB for the largest).
//
// Scan: three passes over blocks of 4096 entries -- block sums, a one-workgroup scan of the block sums, the blocks' own scans with
// their offsets.  Integer arithmetic: the result does not depend on the order of anything.
#include "mpg_internal.h"

#define SCAN_NT 256
#define SCAN_IPT 16
#define SCAN_TILE (SCAN_NT * SCAN_IPT)

// sum of a block's SCAN_TILE entries
__global__ __launch_bounds__(SCAN_NT) void k_scan_block_sums(const int32_t *__restrict__ in, int64_t n, int32_t *__restrict__ bsum) {
  __shared__ int32_t ws[SCAN_NT / 64];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
  int32_t v = 0;
#pragma unroll
  for (int k = 0; k < SCAN_IPT; ++k) {
    const int64_t i = base + (int64_t)k * SCAN_NT + threadIdx.x;   // coalesced: consecutive lanes, consecutive entries
    if (i < n) v += in[i];
  }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) bsum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// exclusive scan of one workgroup's worth of values held one per thread; returns the thread's exclusive prefix, *total = the sum
__device__ __forceinline__ int32_t block_excl_scan(int32_t v, int32_t *total) {
  __shared__ int32_t ws[SCAN_NT / 64 + 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int32_t inc = v;
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  __syncthreads();   // (ws may still be read by a previous call's last readers)
  if (lane == 63) ws[w] = inc;
  __syncthreads();
  int32_t off = 0;
  for (int k = 0; k < w; ++k) off += ws[k];
  *total = ws[0] + ws[1] + ws[2] + ws[3];
  return off + inc - v;
}

// in place over the block sums: bsum[b] <- sum of bsum[0 .. b); one workgroup walks them SCAN_NT at a time
__global__ __launch_bounds__(SCAN_NT) void k_scan_sums(int32_t *__restrict__ bsum, int64_t nb) {
  int32_t carry = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += SCAN_NT) {
    const int64_t i = b0 + threadIdx.x;
    const int32_t v = i < nb ? bsum[i] : 0;
    int32_t total;
    const int32_t ex = block_excl_scan(v, &total);
    if (i < nb) bsum[i] = carry + ex;
    carry += total;
  }
}

// every block scans its own entries (thread t owns the SCAN_IPT consecutive entries t * SCAN_IPT ..: its running sum is local) and adds
// the block's offset
// `in` and `out` may be the same array (mpg_scan_excl_i32 allows in == out): neither is __restrict__ -- every entry a thread writes it has
// read into registers before, but the qualifier would let the compiler assume otherwise-impossible things about the two (round-5 advisor)
__global__ __launch_bounds__(SCAN_NT) void k_scan_apply(const int32_t *in, int64_t n, const int32_t *__restrict__ boff, int32_t *out) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_IPT;
  int32_t v[SCAN_IPT];
  int32_t sum = 0;
#pragma unroll
  for (int k = 0; k < SCAN_IPT; ++k) {
    v[k] = base + k < n ? in[base + k] : 0;
    sum += v[k];
  }
  int32_t total;
  int32_t run = boff[blockIdx.x] + block_excl_scan(sum, &total);
#pragma unroll
  for (int k = 0; k < SCAN_IPT; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
}

// out[i] = in[0] + ... + in[i - 1] for i < n (out[0] = 0); in == out is allowed.  int32 wrap-around is the caller's to check (the
// Stores compare the last entry with a 64-bit sum).
int mpg_scan_excl_i32(const int32_t *in, int32_t *out, int64_t n, hipStream_t s) {
  if (n <= 0) return MPG_SUCCESS;
  const int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  TmpBuf<int32_t> bsum;
  int rc;
  if ((rc = bsum.alloc((size_t)nb, s))) return rc;
  k_scan_block_sums<<<(unsigned)nb, SCAN_NT, 0, s>>>(in, n, bsum.p);
  k_scan_sums<<<1, SCAN_NT, 0, s>>>(bsum.p, nb);
  k_scan_apply<<<(unsigned)nb, SCAN_NT, 0, s>>>(in, n, bsum.p, out);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;   // (bsum goes back to the stream's pool: the next user on this stream is ordered behind k_scan_apply)
}

__global__ __launch_bounds__(256) void k_sum_i32_i64(const int32_t *__restrict__ in, int64_t n, unsigned long long *__restrict__ out) {
  __shared__ long long ws[4];
  long long v = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v += in[i];
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (unsigned long long)(ws[0] + ws[1] + ws[2] + ws[3]));
}

// *out_dev = sum of in[0 .. n) in 64 bits (two's complement: negative entries subtract)
int mpg_sum_i32_i64(const int32_t *in, int64_t n, long long *out_dev, hipStream_t s) {
  MPG_HIP(hipMemsetAsync(out_dev, 0, sizeof(long long), s));
  if (n > 0) k_sum_i32_i64<<<(unsigned)std::min<int64_t>((n + 255) / 256, 1024), 256, 0, s>>>(in, n, (unsigned long long *)out_dev);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

// mpg_init loads this translation unit's code object ahead of its first launch (mpg_api.hip: warm_modules)
const void *mpg_anchor_k_prims() { return (const void *)k_scan_apply; }
