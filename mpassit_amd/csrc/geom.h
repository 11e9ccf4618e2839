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

// Bijective XCD swizzle (cdna_hip_programming.md s5 "XCD swizzle must be bijective"): the hardware deals workgroups b,
// b+8, b+16, ... to the same XCD, so give each of the 8 XCDs one contiguous range of the linear work space -- tiles that
// are neighbours in the work space then share an L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned lin, unsigned n) {
  unsigned q = n / 8, r = n % 8, xcd = lin % 8, k = lin / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// Order of the tiles of ONE field inside the linear work space: row-major tiles would put the tile below (tx, ty) a whole
// tile row (ntx workgroups) later, by when the source cells the two share along their common edge have left the 4 MiB L2
// (C4: 29 tiles x 160 KB in between).  Bands of `band` tile rows walked column by column make vertical neighbours
// consecutive and horizontal neighbours `band` apart; only the band's outer edges are fetched twice.  Bijective on
// [0, ntx * nty); band <= 1 is the row-major order.
__device__ __forceinline__ unsigned band_order(unsigned tl, unsigned ntx, unsigned nty, unsigned band) {
  if (band <= 1) return tl;
  const unsigned per = band * ntx, b = tl / per, r = tl - b * per;
  const unsigned y0 = b * band, bh = min(band, nty - y0);
  const unsigned tx = r / bh, ty = y0 + (r - tx * bh);
  return ty * ntx + tx;
}
