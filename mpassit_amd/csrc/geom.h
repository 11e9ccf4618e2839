// Device-side spherical geometry shared by the weight-generation kernels (float64 throughout).
// Semantics follow SURVEY.md Appendix A (ESMF reference manual); the CPU oracle restates the same
// mathematics independently in oracle/mpassit_oracle.c.
#pragma once
#include <hip/hip_runtime.h>

#define MPG_TOL 1e-10  // "inside" tolerance on barycentric / parametric coordinates (App. A2)

struct dv3 {
  double x, y, z;
};
__device__ __forceinline__ dv3 mk3(double x, double y, double z) { return dv3{x, y, z}; }
__device__ __forceinline__ dv3 operator-(dv3 a, dv3 b) { return dv3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ dv3 operator+(dv3 a, dv3 b) { return dv3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ dv3 operator*(dv3 a, double s) { return dv3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ double dot3(dv3 a, dv3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ dv3 cross3(dv3 a, dv3 b) {
  return dv3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ dv3 ld3(const double *x, const double *y, const double *z, int64_t i) {
  return dv3{x[i], y[i], z[i]};
}
// det[p,b,c] through differences from p (well conditioned for km-scale triangles on the unit sphere)
__device__ __forceinline__ double det3_from(dv3 p, dv3 b, dv3 c) { return dot3(p, cross3(b - p, c - p)); }

// A2: barycentric weights of P in planar triangle ABC seen from the origin.  Returns true when inside.
__device__ __forceinline__ bool tri_weights(dv3 P, dv3 A, dv3 B, dv3 C, double tol, double *w) {
  dv3 a = A - P, b = B - P, c = C - P;
  double dA = dot3(P, cross3(b, c)), dB = dot3(P, cross3(c, a)), dC = dot3(P, cross3(a, b));
  double S = dA + dB + dC;
  if (!(S > 0.0)) return false;
  w[0] = dA / S;
  w[1] = dB / S;
  w[2] = dC / S;
  return w[0] >= -tol && w[1] >= -tol && w[2] >= -tol;
}

// The other reading of "straight cell edges on a sphere": P is dropped onto the triangle's plane along the plane's NORMAL
// (ESMF_LINETYPE_CART taken literally: the element is a flat triangle of 3-D space and the point is located in its local
// coordinates) instead of along the ray from the sphere's centre as above, and takes the barycentric coordinates of its
// foot.  The two differ by O(h^2) of the triangle size; selected per Store with mpg_tune("bilinear_linetype", 1).
__device__ __forceinline__ bool tri_weights_normal(dv3 P, dv3 A, dv3 B, dv3 C, double tol, double *w) {
  dv3 n = cross3(B - A, C - A);
  double nn = dot3(n, n);
  if (!(nn > 0.0) || !(dot3(n, P) > 0.0)) return false;
  double d = dot3(P - A, n) / nn;
  dv3 F = P - n * d;
  dv3 a = A - F, b = B - F, c = C - F;
  double dA = dot3(n, cross3(b, c)), dB = dot3(n, cross3(c, a)), dC = dot3(n, cross3(a, b));
  double S = dA + dB + dC;
  if (!(S > 0.0)) return false;
  w[0] = dA / S;
  w[1] = dB / S;
  w[2] = dC / S;
  return w[0] >= -tol && w[1] >= -tol && w[2] >= -tol;
}

// A6: squared chord distance, evaluated as ((dx^2 + dy^2) + dz^2) WITHOUT fma so that host (oracle) and
// device agree bit for bit and box lower bounds stay monotone.
__device__ __forceinline__ double dist2_nofma(double px, double py, double pz, double cx, double cy, double cz) {
#pragma clang fp contract(off)
  double dx = px - cx, dy = py - cy, dz = pz - cz;
  double a = dx * dx, b = dy * dy, c = dz * dz;
  return (a + b) + c;
}
__device__ __forceinline__ double boxdist2_nofma(double px, double py, double pz, const double *bx) {
#pragma clang fp contract(off)
  double dx = fmax(fmax(bx[0] - px, px - bx[3]), 0.0);
  double dy = fmax(fmax(bx[1] - py, py - bx[4]), 0.0);
  double dz = fmax(fmax(bx[2] - pz, pz - bx[5]), 0.0);
  double a = dx * dx, b = dy * dy, c = dz * dz;
  return (a + b) + c;
}

// A7: the 3-term weighted sum of every bilinear Regrid kernel, with the FMA pattern pinned so that all kernel
// variants (cell-fast, level-fast, typed) produce bit-identical results
// Pad (index units) of the index-space box of a figure E index units across whose vertices reach |z| = zmax on the unit sphere
// (k_target_grid.hip: mpg_grid_box_pad_coef).  coef: Lambert, per E^2.  latlon > 0: a lat-lon grid with cells of `latlon` radians
// -- the image of a great circle bends by tan(lat) * E_i * E_j * delta / 4 in i and sin cos(lat) * E_i^2 * delta / 8 in j; twice
// their sum, with the latitude of the figure itself, so that figures at low latitudes get a tight box and figures near the poles
// a wide one instead of no box at all.  0.05 covers the float32 indices.
__device__ __forceinline__ float mpg_box_pad(float E, float coef, float latlon, double zmax) {
  if (latlon > 0.f) {
    const float z = fminf((float)zmax, 0.99999f);
    const float tanl = z / sqrtf(1.f - z * z);
    return 0.05f + E * E * latlon * (0.5f * tanl + 0.125f);
  }
  return 0.05f + coef * E * E;
}

__device__ __forceinline__ double wsum3(double w0, double a, double w1, double b, double w2, double e) {
  return fma(w2, e, fma(w1, b, w0 * a));
}

// A5: signed spherical triangle area (Van Oosterom-Strackee), difference form
__device__ __forceinline__ double sph_tri_area(dv3 a, dv3 b, dv3 c) {
  double num = det3_from(a, b, c);
  double den = 1.0 + dot3(a, b) + dot3(b, c) + dot3(c, a);
  // km-scale triangles: x = num/den ~ 1e-7, where atan's series x - x^3/3 + x^5/5 - x^7/7 is exact to the last bit
  // (next term < 1e-28 relative at |x| < 1e-3) and costs one division instead of an atan2
  if (den > 0.0 && fabs(num) < 1e-3 * den) {
    double x = num / den, x2 = x * x;
    return 2.0 * x * (1.0 - x2 * (1.0 / 3.0 - x2 * (1.0 / 5.0 - x2 * (1.0 / 7.0))));
  }
  return 2.0 * atan2(num, den);
}

// ---- streaming stores of results ---------------------------------------------------------------------------------------
// Results are written once and never read by the kernel, so a wavefront's run of 64 consecutive result elements goes out
// non-temporal (2-6 % on the Regrid kernels: the lines do not displace the source rows the kernel lives on in L2) -- where it
// covers whole 128-byte lines.  A run that STARTS inside a line leaves a partial line at either end, whose other part a
// neighbouring workgroup writes a little later: non-temporal, the first part has left L2 by then and HBM sees two masked writes
// per line; write-back, the parts meet in L2.  Whether runs start on a line is a property of the level's PLANE: tiles start at
// multiples of 64 points of the plane, so level k's runs are aligned iff its plane is -- and plane k of a [nlev][ny][nx] result
// starts k * ny * nx * sizeof(T) bytes after plane 0: on a line for every k only when ny * nx is a multiple of 32 (float32) / 16
// (float64) points.  The headline grid (1800 x 1060) is -- which hid all this for five rounds; its staggers (1801 x 1060, 1800 x
// 1061), a rank's row block of it (1800 x 133), HRRR's own 1799 x 1059 and most grids a user brings are not: as shipped until
// round 6a they cost 14 % (float64 cell-fast) to 29 % (float32 file order) of the kernel (profiles/r06_plane_alignment.md).
//   stream_store_lane(v, p, lane_bytes)   per LANE, the general form (below): whole lines non-temporal, the run's two end lines
//                                         write-back; no branch, safe inside pipelined loops
//   stream_nt(plane), stream_store(v, p, nt)   per LEVEL (wave-uniform: this level's plane starts on a line).  ONLY for stores in a
//                             kernel's final phase: a uniform branch around stores inside a loop that also loads makes the compiler
//                             wait for every outstanding store (one in-order counter on gfx950; measured: 10 % of k_wind_destagger)
// The empty asm statements keep LLVM from merging the two branches into ONE plain store (it sinks / hoists stores that differ
// only in their !nontemporal hint and drops the hint: found in the ISA, round 6).
// -DMPG_STREAM_STORE_MODE=1 / 2 (A/B builds, mpassit_amd.build.build_alt): everything plain / non-temporal (2 = rounds 2-6a).
#ifndef MPG_STREAM_STORE_MODE
#define MPG_STREAM_STORE_MODE 0
#endif
__device__ __forceinline__ bool stream_nt(const void *plane, int mode = MPG_STREAM_STORE_MODE) {
  if (mode) return mode == 2;
  return __builtin_amdgcn_readfirstlane((int)((uintptr_t)plane & 127u)) == 0;   // all lanes of a workgroup work on one level of one field
}
// Per LANE (round 6, the general form): a lane whose 128-byte line lies wholly inside its wavefront's run of 64 consecutive
// elements stores non-temporal, the lanes on the partial lines at the run's two ends store write-back, so that the other part --
// the neighbouring run's -- meets them in L2.  No wave-uniform branch: both stores are straight-line code under complementary
// lane masks, so the compiler's count of what is in flight stays static and the form is safe inside pipelined level loops
// (same s_waitcnt pattern in the ISA; k_apply3_cfu float64 on planes 8 / 24 / 104 bytes off a line: 0.57 -> 0.62-0.64 of the
// peak, planes on a line or 64 bytes off unchanged, profiles/r06_plane_alignment.md).  lane_bytes = (lane of the run) * sizeof(T);
// a run cut short by the end of a grid row may send one partial line non-temporal: rare, and only a matter of speed.
template <typename T>
__device__ __forceinline__ bool stream_lane_full(const T *addr, unsigned lane_bytes) {
  const unsigned a = (unsigned)(uintptr_t)addr & 127u;
  return a <= lane_bytes && lane_bytes - a <= 64u * (unsigned)sizeof(T) - 128u;
}
template <typename T>
__device__ __forceinline__ void stream_store(T v, T *addr, bool nt) {
  if (nt) {
    asm volatile("" ::: "memory");
    __builtin_nontemporal_store(v, addr);
    asm volatile("" ::: "memory");
  } else {
    *addr = v;
  }
}
// all_nt (wave-uniform, a run-time A/B knob): OR-ed into the lane predicate, no branch -- the write-back store then runs with no lane
template <typename T>
__device__ __forceinline__ void stream_store_lane(T v, T *addr, unsigned lane_bytes, bool all_nt = false) {
  stream_store(v, addr, MPG_STREAM_STORE_MODE ? MPG_STREAM_STORE_MODE == 2 : (all_nt | stream_lane_full(addr, lane_bytes)));
}

// Bijective XCD swizzle (cdna_hip_programming.md s5 "XCD swizzle must be bijective"): the hardware deals workgroups b,
// b+8, b+16, ... to the same XCD, so give each of the 8 XCDs one contiguous range of the linear work space -- tiles that
// are neighbours in the work space then share an L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned lin, unsigned n) {
  unsigned q = n / 8, r = n % 8, xcd = lin % 8, k = lin / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// Order of the (field, tile) work items of a bundle Regrid.  band == 0: field-major (all tiles of field 0, then field 1 ...).
// band > 0: the tiles are taken in bands of `band`, and ALL FIELDS of a band come before the next band -- the band's
// indices, weights and tile lists (the same for every field: 30-36 B per point) are then re-read from L2 by the next
// field instead of from HBM a whole field later ("field_band" knob; measurements in profiles/r04_lf_experiments.txt).
__device__ __forceinline__ void band_map(unsigned lin, unsigned ntile, unsigned nf, unsigned band, unsigned &tile, int &f) {
  if (band == 0) {
    tile = lin % ntile;
    f = (int)(lin / ntile);
    return;
  }
  const unsigned nb = ntile / band, rem = ntile - nb * band, full = nb * band * nf;
  if (lin < full) {
    const unsigned per = band * nf, b = lin / per, r = lin - b * per;
    f = (int)(r / band);
    tile = b * band + (r - (unsigned)f * band);
  } else {
    const unsigned r = lin - full;
    f = (int)(r / rem);
    tile = nb * band + (r - (unsigned)f * rem);
  }
}

// ---- element access of the fused ingest / egress kernels ------------------------------------------------------------
// A NetCDF classic variable is big-endian (the files MPASSIT reads, input_data.F90:630, and writes, write_data.F90:779,
// when they are CDF-1/2/5): the typed Regrid and the post-op kernels take and produce such values as they are stored, so
// that no separate byte-swap pass touches the data (round 2: k_bswap was the largest item of a cold configuration-4 job,
// one extra read + write of every ingested and emitted byte).  The swap is one v_perm_b32 per 32-bit word with a
// wave-uniform selector: identity or byte reversal, so one kernel serves any mix of byte orders on its two sides.
struct Swz {
  uint32_t s32;            // 32-bit element: selector over {0, word}
  uint32_t lo64, hi64;     // 64-bit element: selectors over {hi, lo} for the new low / high word
};
__device__ __forceinline__ Swz make_swz(int big_endian) {
  Swz z;
  z.s32 = big_endian ? 0x00010203u : 0x03020100u;
  z.lo64 = big_endian ? 0x04050607u : 0x03020100u;
  z.hi64 = big_endian ? 0x00010203u : 0x07060504u;
  return z;
}
template <bool SWZ>
__device__ __forceinline__ float swz(float v, const Swz &z) {
  if constexpr (!SWZ) return v;
  return __uint_as_float(__builtin_amdgcn_perm(0u, __float_as_uint(v), z.s32));
}
template <bool SWZ>
__device__ __forceinline__ double swz(double v, const Swz &z) {
  if constexpr (!SWZ) return v;
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)b, hi = (uint32_t)(b >> 32);
  const uint32_t nlo = __builtin_amdgcn_perm(hi, lo, z.lo64), nhi = __builtin_amdgcn_perm(hi, lo, z.hi64);
  return __longlong_as_double((long long)(((unsigned long long)nhi << 32) | nlo));
}

// ---- buffer addressing (scalar base + 32-bit lane offset) -------------------------------------------------------------
// A raw buffer descriptor gives the staged kernels two things the flat form cannot: no 64-bit address arithmetic per
// access (the base and a per-chunk / per-level offset live in scalar registers) and hardware range checking -- a lane
// whose offset is >= the descriptor's size loads 0 and its store is dropped, so lanes without a target point need no
// branch around their stores (a branch there makes the compiler wait for ALL outstanding stores before the next chunk's
// loads can be consumed: loads and stores share one counter on gfx950).  Sizes are bytes and must stay below 4 GB.
typedef __amdgpu_buffer_rsrc_t BufRsrc;
typedef unsigned buf_u32x2 __attribute__((ext_vector_type(2)));
#define MPG_BUF_NONE 0xFFFFFFFFu   // lane offset of a lane that must not touch memory
__device__ __forceinline__ BufRsrc buf_rsrc(const void *base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
// -DMPG_SRC_NT (A/B builds only, mpassit_amd.build.build_alt): the staged kernels' source loads carry the "nt" bit -- every cell is
// loaded once per tile, L1 has nothing to give them.  Measured 16-55 % SLOWER (profiles/r06_src_nt_loads.txt: the bit also makes the lines evict-first
// in L2, and a row's next 64-byte chunk and the neighbour tile's ring are then fetched again): not in the product build.
#ifdef MPG_SRC_NT
#define MPG_SRC_AUX 2
#define MPG_LDG(p) __builtin_nontemporal_load(p)
#else
#define MPG_SRC_AUX 0
#define MPG_LDG(p) (*(p))
#endif
__device__ __forceinline__ void buf_load(float &v, BufRsrc r, uint32_t lane_off, uint32_t wave_off) {
  v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_off, (int)wave_off, MPG_SRC_AUX));
}
__device__ __forceinline__ void buf_load(double &v, BufRsrc r, uint32_t lane_off, uint32_t wave_off) {
  v = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)lane_off, (int)wave_off, MPG_SRC_AUX));
}
// non-temporal (aux = 2: the "nt" bit of gfx94x / gfx950)
__device__ __forceinline__ void buf_store_nt(float v, BufRsrc r, uint32_t lane_off) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)lane_off, 0, 2);
}
__device__ __forceinline__ void buf_store_nt(double v, BufRsrc r, uint32_t lane_off) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(buf_u32x2, v), r, (int)lane_off, 0, 2);
}
// plain (write-back), and the choice between the two at COMPILE time (a kernel whose stores sit in a loop that also loads picks per
// instantiation: see stream_store above)
__device__ __forceinline__ void buf_store_wb(float v, BufRsrc r, uint32_t lane_off) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)lane_off, 0, 0);
}
__device__ __forceinline__ void buf_store_wb(double v, BufRsrc r, uint32_t lane_off) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(buf_u32x2, v), r, (int)lane_off, 0, 0);
}
template <bool NT, typename T>
__device__ __forceinline__ void buf_store_pick(T v, BufRsrc r, uint32_t lane_off) {
  if constexpr (NT) buf_store_nt(v, r, lane_off);
  else buf_store_wb(v, r, lane_off);
}
// stream_store_lane for buffer addressing: `plane` = the descriptor's base, lane_off = the lane's byte offset inside it (MPG_BUF_NONE:
// no store either way).  Two stores under complementary lane masks, distinct cache-policy immediates: nothing for the compiler to merge.
template <typename T>
__device__ __forceinline__ void buf_store_lane(T v, BufRsrc r, const void *plane, uint32_t lane_off, unsigned lane_bytes) {
  const unsigned a = ((unsigned)(uintptr_t)plane + lane_off) & 127u;
  const bool full = MPG_STREAM_STORE_MODE ? MPG_STREAM_STORE_MODE == 2 : (a <= lane_bytes && lane_bytes - a <= 64u * (unsigned)sizeof(T) - 128u);
  if (full) buf_store_nt(v, r, lane_off);
  else buf_store_wb(v, r, lane_off);
}
