/*
 * mpassit_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 * Plain-C float64 CPU restatement of the interpolation arithmetic that
 * LarissaReames-NOAA/MPASSIT delegates to ESMF (interp.F90:123-447) plus the small
 * in-house pieces on the same path (rotate_winds_cgrid interp.F90:689-749,
 * get_rotang model_grid.F90:2450-2507, target-grid coordinates
 * model_grid.F90:2188-2219 / llxy_module.F90:166-216 /
 * module_map_utils.F90:1083-1290,1398-1428, para_range model_grid.F90:2428-2441).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (mpassit_amd/) never links, imports or calls it.
 *
 * PARITY STATUS: **parity unpinned** at the ESMF boundary.  The arithmetic lives in
 * ESMF (third party, >= 8.3.0 per CMakeLists.txt:48; author used 8.6.0,
 * modulefiles/build.jet.intel.lua:30), which is neither vendored in the reference tree
 * nor installed here, and the reference ships no tests/golden vectors.  This file
 * restates the published ESMF semantics summarised in SURVEY.md Appendix A.
 * What IS pinned: the target-grid projection (orc_lc_*, orc_xytoll) reproduces the
 * compiled-reference golden values recorded in SURVEY.md Appendix E
 * (tests/golden/projection_lc.json), and every routine has analytic known-answer tests.
 * Pinned to the MATHEMATICS (not to ESMF, not to the reference's binary) by independent high-precision goldens with
 * committed generators: the four projections at 4 680 points (tests/golden/make_projection_goldens.py, mpmath 40 digits)
 * and the bilinear / conservative / quad-bilinear weight formulas at 50 digits along a different route
 * (tests/golden/make_weight_goldens.py).
 *
 * Search structures here (3-D box hash) deliberately differ from the GPU ones
 * (target pyramid rasteriser / Morton BVH) so that agreement is a real cross-check.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define ORC_TOL 1e-10          /* "inside" tolerance on barycentric / parametric coords (App. A2) */
#define ORC_PI 3.141592653589793 /* constants_module.F90:8 */
#define ORC_RAD_PER_DEG (ORC_PI / 180.)
#define ORC_DEG_PER_RAD (180. / ORC_PI)
#define ORC_EARTH_RADIUS_M 6370000. /* constants_module.F90:25 */

typedef struct { double x, y, z; } v3;
static inline v3 v3sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline v3 v3add(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static inline v3 v3scale(v3 a, double s) { v3 r = {a.x * s, a.y * s, a.z * s}; return r; }
static inline double v3dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3cross(v3 a, v3 b) {
  v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
  return r;
}
static inline v3 v3load(const double *p, int64_t i) { v3 r = {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; return r; }
static inline v3 v3norm(v3 a) { double n = sqrt(v3dot(a, a)); return v3scale(a, 1.0 / n); }
/* det[a,b,c] evaluated through differences from a (well conditioned for small triangles) */
static inline double det3_from(v3 p, v3 b, v3 c) { return v3dot(p, v3cross(v3sub(b, p), v3sub(c, p))); }

/* ------------------------------------------------------------------------------------------
 * A1. Coordinates.  model_grid.F90:450-454,464-468: lon(rad)*180/PI, >180 -> -360; lat*180/PI.
 * PI there is 4*atan(1) (model_grid.F90:280).  ESMF then maps degrees to the unit sphere.
 * ---------------------------------------------------------------------------------------- */
void orc_mesh_coords_deg(int64_t n, const double *lon_rad, const double *lat_rad, double *lon_deg,
                         double *lat_deg) {
  const double PI = 4.0 * atan(1.0);
  for (int64_t i = 0; i < n; ++i) {
    double lo = lon_rad[i] * 180.0 / PI;
    if (lo > 180.0) lo -= 360.0;
    lon_deg[i] = lo;
    lat_deg[i] = lat_rad[i] * 180.0 / PI;
  }
}

void orc_lonlat_deg_to_xyz(int64_t n, const double *lon_deg, const double *lat_deg, double *xyz) {
  const double d2r = ORC_PI / 180.0;
  for (int64_t i = 0; i < n; ++i) {
    double lo = lon_deg[i] * d2r, la = lat_deg[i] * d2r;
    double cl = cos(la);
    xyz[3 * i] = cl * cos(lo);
    xyz[3 * i + 1] = cl * sin(lo);
    xyz[3 * i + 2] = sin(la);
  }
}

/* ------------------------------------------------------------------------------------------
 * A2. Dual mesh.  One dual element per MPAS vertex = the cells listing that vertex in
 * verticesOnCell (1-based, 0 = pad; model_grid.F90:448,474-485).  Vertices touched by != 3
 * cells give no triangle.  tri[3*v+k] = 0-based cell id, or -1.  Oriented so det[A,B,C] > 0.
 * Returns the number of valid triangles.
 * ---------------------------------------------------------------------------------------- */
int64_t orc_dual_triangles(int64_t nCells, int64_t nVertices, int maxEdges, const int32_t *voc,
                           const double *cell_xyz, int32_t *tri) {
  int32_t *cnt = (int32_t *)calloc((size_t)nVertices, sizeof(int32_t));
  for (int64_t v = 0; v < 3 * nVertices; ++v) tri[v] = -1;
  for (int64_t c = 0; c < nCells; ++c)
    for (int j = 0; j < maxEdges; ++j) {
      int32_t v = voc[c * maxEdges + j];
      if (v <= 0 || v > nVertices) continue;
      v -= 1;
      if (cnt[v] < 3) tri[3 * v + cnt[v]] = (int32_t)c;
      cnt[v]++;
    }
  int64_t nvalid = 0;
  for (int64_t v = 0; v < nVertices; ++v) {
    if (cnt[v] != 3) { tri[3 * v] = tri[3 * v + 1] = tri[3 * v + 2] = -1; continue; }
    v3 A = v3load(cell_xyz, tri[3 * v]), B = v3load(cell_xyz, tri[3 * v + 1]), C = v3load(cell_xyz, tri[3 * v + 2]);
    double d = det3_from(A, B, C);
    if (d < 0) { int32_t t = tri[3 * v + 1]; tri[3 * v + 1] = tri[3 * v + 2]; tri[3 * v + 2] = t; }
    if (d == 0) { tri[3 * v] = tri[3 * v + 1] = tri[3 * v + 2] = -1; continue; }
    nvalid++;
  }
  free(cnt);
  return nvalid;
}

/* ------------------------------------------------------------------------------------------
 * A3. Bilinear, source on nodes (vorticity; interp.F90:350-366, input_data.F90:1116-1123).  The original
 * mesh is used: Voronoi polygons with values at their corners.  ESMF triangulates polygons with more than 4
 * sides internally in an undocumented order [ESMF-doc] => IMPLEMENTATION-DEFINED; this restatement (and the
 * GPU) use the fan from the first listed vertex: triangle k of cell c = (v0, v_{k+1}, v_{k+2}),
 * id = c*(maxEdges-2)+k, lowest id wins on shared edges.  ftri[3*id+..] = 0-based vertex ids or -1.
 * orc_set_fan_origin(o): the fan's apex is the listed vertex number o mod n instead (o = -1: the LAST listed vertex -- what an
 * ear-clipping loop that always cuts the first ear of a convex polygon produces), the others follow in listed order from there:
 * the kernels' "node_fan_origin" knob, so that a site's ESMF comparison can say which split its library makes.
 * ---------------------------------------------------------------------------------------- */
static int g_orc_fan_origin = 0;
void orc_set_fan_origin(int o) { g_orc_fan_origin = o; }
int64_t orc_fan_triangles(int64_t nCells, int maxEdges, const int32_t *voc, const double *vert_xyz, int32_t *ftri) {
  int nf = maxEdges - 2;
  int64_t nvalid = 0;
  for (int64_t t = 0; t < 3 * nCells * nf; ++t) ftri[t] = -1;
  for (int64_t c = 0; c < nCells; ++c) {
    int32_t v[64]; int n = 0;
    for (int j = 0; j < maxEdges && n < 64; ++j) { int32_t x = voc[c * maxEdges + j]; if (x > 0) v[n++] = x - 1; }
    if (n == 0) continue;
    const int o = ((g_orc_fan_origin % n) + n) % n;
    for (int k = 0; k + 2 < n; ++k) {
      int64_t t = c * nf + k;
      int32_t a = v[o], b = v[(o + k + 1) % n], d = v[(o + k + 2) % n];
      double det = det3_from(v3load(vert_xyz, a), v3load(vert_xyz, b), v3load(vert_xyz, d));
      if (det == 0) continue;
      if (det < 0) { int32_t x = b; b = d; d = x; }
      ftri[3 * t] = a; ftri[3 * t + 1] = b; ftri[3 * t + 2] = d;
      nvalid++;
    }
  }
  return nvalid;
}

/* ------------------------------------------------------------------------------------------
 * 3-D box hash: every item registers in each grid cell of [-1,1]^3 its (inflated) AABB touches.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  double g;       /* cell size */
  int G;          /* cells per axis */
  uint32_t M;     /* #buckets (power of two) */
  int64_t *start; /* [M+1] */
  int32_t *items;
  int32_t *big;   /* items too large for the grid: scanned by every query */
  int64_t nbig;
} boxhash;

static inline int bh_cell(const boxhash *h, double x) {
  int i = (int)floor((x + 1.0) / h->g);
  if (i < 0) i = 0;
  if (i >= h->G) i = h->G - 1;
  return i;
}
static inline uint32_t bh_hash(const boxhash *h, int ix, int iy, int iz) {
  uint64_t k = ((uint64_t)ix * 73856093u) ^ ((uint64_t)iy * 19349663u) ^ ((uint64_t)iz * 83492791u);
  k ^= k >> 29; k *= 0x9E3779B97F4A7C15ull; k ^= k >> 32;
  return (uint32_t)k & (h->M - 1);
}
#define BH_MAXCELLS 512
static void bh_build(boxhash *h, int64_t n, const double *lo, const double *hi, const uint8_t *valid, double g) {
  if (g < 1e-5) g = 1e-5;
  h->g = g; h->G = (int)ceil(2.0 / g); if (h->G < 1) h->G = 1;
  uint64_t want = 1; while (want < (uint64_t)(4 * n + 16)) want <<= 1; if (want > (1ull << 28)) want = 1ull << 28;
  h->M = (uint32_t)want;
  h->start = (int64_t *)calloc((size_t)h->M + 2, sizeof(int64_t));
  h->nbig = 0; h->big = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
  for (int pass = 0; pass < 2; ++pass) {
    for (int64_t i = 0; i < n; ++i) {
      if (valid && !valid[i]) continue;
      int a0 = bh_cell(h, lo[3 * i]), a1 = bh_cell(h, hi[3 * i]);
      int b0 = bh_cell(h, lo[3 * i + 1]), b1 = bh_cell(h, hi[3 * i + 1]);
      int c0 = bh_cell(h, lo[3 * i + 2]), c1 = bh_cell(h, hi[3 * i + 2]);
      int64_t nc = (int64_t)(a1 - a0 + 1) * (b1 - b0 + 1) * (c1 - c0 + 1);
      if (nc > BH_MAXCELLS) { if (pass == 0) h->big[h->nbig++] = (int32_t)i; continue; }
      for (int a = a0; a <= a1; ++a) for (int b = b0; b <= b1; ++b) for (int c = c0; c <= c1; ++c) {
        uint32_t k = bh_hash(h, a, b, c);
        if (pass == 0) h->start[k + 2]++;
        else h->items[h->start[k + 1]++] = (int32_t)i;
      }
    }
    if (pass == 0) {
      for (uint32_t k = 0; k < h->M; ++k) h->start[k + 2] += h->start[k + 1];
      h->items = (int32_t *)malloc(sizeof(int32_t) * (size_t)(h->start[h->M + 1] + 1));
    }
  }
  /* after pass 1, start[k+1] == end of bucket k == begin of bucket k+1; start[k] = begin(k) */
}
static void bh_free(boxhash *h) { free(h->start); free(h->items); free(h->big); }

/* ------------------------------------------------------------------------------------------
 * A2. Bilinear, source on elements: containing Delaunay (dual) triangle + 3 weights.
 *   d_A=det[P,B,C], d_B=det[A,P,C], d_C=det[A,B,P], S=sum, w=d/S; inside iff all w >= -tol, S>0.
 * Lowest triangle (= vertex) id wins on shared edges.  Unmapped: idx=-1, w=0
 * (unmappedaction=IGNORE + zero-filled destination => 0.0; interp.F90:127).
 * ---------------------------------------------------------------------------------------- */
static inline int tri_weights(v3 P, v3 A, v3 B, v3 C, double tol, double *w) {
  v3 a = v3sub(A, P), b = v3sub(B, P), c = v3sub(C, P);
  double dA = v3dot(P, v3cross(b, c)), dB = v3dot(P, v3cross(c, a)), dC = v3dot(P, v3cross(a, b));
  double S = dA + dB + dC;
  if (!(S > 0)) return 0;
  w[0] = dA / S; w[1] = dB / S; w[2] = dC / S;
  return (w[0] >= -tol && w[1] >= -tol && w[2] >= -tol);
}

/* The other reading of "straight cell edges on a sphere" (ESMF_LINETYPE_CART taken literally): the destination point is
 * dropped onto the triangle's plane along the plane's NORMAL instead of along the ray from the sphere's centre, and takes
 * the barycentric coordinates of its foot.  The two differ by O(h^2) of the triangle size h; orc_set_linetype(1) selects
 * this form for orc_bilinear_weights (Mesh -> Grid only) so that the difference can be measured. */
static int g_orc_linetype = 0;
void orc_set_linetype(int v) { g_orc_linetype = v; }
/* A4: the tolerance within which a stagger point counts as inside a quad of CENTER points (parametric coordinates); the kernels'
 * "grid_inside_tol_exp" knob: 10^-e, default e = 10 */
static double g_orc_grid_tol = ORC_TOL;
void orc_set_grid_tol(double t) { g_orc_grid_tol = t; }
static inline int tri_weights_normal(v3 P, v3 A, v3 B, v3 C, double tol, double *w) {
  v3 n = v3cross(v3sub(B, A), v3sub(C, A));
  double nn = v3dot(n, n);
  if (!(nn > 0) || !(v3dot(n, P) > 0)) return 0;
  double d = v3dot(v3sub(P, A), n) / nn;
  v3 F = {P.x - d * n.x, P.y - d * n.y, P.z - d * n.z};
  v3 a = v3sub(A, F), b = v3sub(B, F), c = v3sub(C, F);
  double dA = v3dot(n, v3cross(b, c)), dB = v3dot(n, v3cross(c, a)), dC = v3dot(n, v3cross(a, b));
  double S = dA + dB + dC;
  if (!(S > 0)) return 0;
  w[0] = dA / S; w[1] = dB / S; w[2] = dC / S;
  return (w[0] >= -tol && w[1] >= -tol && w[2] >= -tol);
}
static inline int tri_weights_lt(v3 P, v3 A, v3 B, v3 C, double tol, double *w) {
  return g_orc_linetype ? tri_weights_normal(P, A, B, C, tol, w) : tri_weights(P, A, B, C, tol, w);
}

static double tri_maxedge(v3 A, v3 B, v3 C) {
  v3 ab = v3sub(B, A), bc = v3sub(C, B), ca = v3sub(A, C);
  double e = fmax(v3dot(ab, ab), fmax(v3dot(bc, bc), v3dot(ca, ca)));
  return sqrt(e);
}

static int cmp_double(const void *a, const void *b) {
  double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

void orc_bilinear_weights(int64_t nCells, const double *cell_xyz, int64_t nTri, const int32_t *tri,
                          int64_t P, const double *pt_xyz, int32_t *idx, double *w) {
  (void)nCells;
  double *lo = (double *)malloc(sizeof(double) * 3 * (size_t)nTri), *hi = (double *)malloc(sizeof(double) * 3 * (size_t)nTri);
  uint8_t *valid = (uint8_t *)malloc((size_t)nTri);
  double *edges = (double *)malloc(sizeof(double) * (size_t)(nTri + 1));
  int64_t ne = 0;
  for (int64_t t = 0; t < nTri; ++t) {
    valid[t] = tri[3 * t] >= 0;
    if (!valid[t]) continue;
    v3 A = v3load(cell_xyz, tri[3 * t]), B = v3load(cell_xyz, tri[3 * t + 1]), C = v3load(cell_xyz, tri[3 * t + 2]);
    double e = tri_maxedge(A, B, C), pad = 0.5 * e * e + 1e-9;
    edges[ne++] = e;
    lo[3 * t] = fmin(A.x, fmin(B.x, C.x)) - pad; hi[3 * t] = fmax(A.x, fmax(B.x, C.x)) + pad;
    lo[3 * t + 1] = fmin(A.y, fmin(B.y, C.y)) - pad; hi[3 * t + 1] = fmax(A.y, fmax(B.y, C.y)) + pad;
    lo[3 * t + 2] = fmin(A.z, fmin(B.z, C.z)) - pad; hi[3 * t + 2] = fmax(A.z, fmax(B.z, C.z)) + pad;
  }
  double g = 0.05;
  if (ne > 0) { qsort(edges, (size_t)ne, sizeof(double), cmp_double); g = 1.5 * edges[ne / 2]; }
  boxhash h; bh_build(&h, nTri, lo, hi, valid, g);
  for (int64_t p = 0; p < P; ++p) {
    v3 Pt = v3load(pt_xyz, p);
    int32_t best = -1; double bw[3] = {0, 0, 0}, wt[3];
    uint32_t k = bh_hash(&h, bh_cell(&h, Pt.x), bh_cell(&h, Pt.y), bh_cell(&h, Pt.z));
    for (int64_t q = h.start[k]; q < h.start[k + 1]; ++q) {
      int32_t t = h.items[q];
      if (best >= 0 && t >= best) continue;
      if (Pt.x < lo[3 * t] || Pt.x > hi[3 * t] || Pt.y < lo[3 * t + 1] || Pt.y > hi[3 * t + 1] || Pt.z < lo[3 * t + 2] || Pt.z > hi[3 * t + 2]) continue;
      if (tri_weights_lt(Pt, v3load(cell_xyz, tri[3 * t]), v3load(cell_xyz, tri[3 * t + 1]), v3load(cell_xyz, tri[3 * t + 2]), ORC_TOL, wt)) {
        best = t; bw[0] = wt[0]; bw[1] = wt[1]; bw[2] = wt[2];
      }
    }
    for (int64_t q = 0; q < h.nbig; ++q) {
      int32_t t = h.big[q];
      if (best >= 0 && t >= best) continue;
      if (tri_weights_lt(Pt, v3load(cell_xyz, tri[3 * t]), v3load(cell_xyz, tri[3 * t + 1]), v3load(cell_xyz, tri[3 * t + 2]), ORC_TOL, wt)) {
        best = t; bw[0] = wt[0]; bw[1] = wt[1]; bw[2] = wt[2];
      }
    }
    if (best >= 0) {
      idx[3 * p] = tri[3 * best]; idx[3 * p + 1] = tri[3 * best + 1]; idx[3 * p + 2] = tri[3 * best + 2];
      w[3 * p] = bw[0]; w[3 * p + 1] = bw[1]; w[3 * p + 2] = bw[2];
    } else {
      idx[3 * p] = idx[3 * p + 1] = idx[3 * p + 2] = -1;
      w[3 * p] = w[3 * p + 1] = w[3 * p + 2] = 0.0;
    }
  }
  bh_free(&h); free(lo); free(hi); free(valid); free(edges);
}

/* ------------------------------------------------------------------------------------------
 * A6. Nearest source-to-destination: argmin over ALL cell centres of the 3-D chord distance,
 * ties -> lowest cell id.  Every destination point is mapped.
 * d2 is evaluated as ((px-cx)^2 + (py-cy)^2) + (pz-cz)^2, no FMA (the GPU kernel matches it).
 * ---------------------------------------------------------------------------------------- */
static inline double dist2(v3 p, v3 c) {
  double dx = p.x - c.x, dy = p.y - c.y, dz = p.z - c.z;
  return (dx * dx + dy * dy) + dz * dz;
}

void orc_nearest_brute(int64_t nCells, const double *cell_xyz, int64_t P, const double *pt_xyz, int32_t *idx) {
  for (int64_t p = 0; p < P; ++p) {
    v3 Pt = v3load(pt_xyz, p);
    double best = INFINITY; int32_t bi = -1;
    for (int64_t c = 0; c < nCells; ++c) {
      double d = dist2(Pt, v3load(cell_xyz, c));
      if (d < best) { best = d; bi = (int32_t)c; }
    }
    idx[p] = bi;
  }
}

void orc_nearest(int64_t nCells, const double *cell_xyz, int64_t P, const double *pt_xyz, int32_t *idx) {
  if (nCells <= 2048) { orc_nearest_brute(nCells, cell_xyz, P, pt_xyz, idx); return; }
  /* grid with ~2 sites per occupied cell: surface area 4pi over nCells sites */
  double g = sqrt(4.0 * ORC_PI / (double)nCells) * 1.5;
  boxhash h; bh_build(&h, nCells, cell_xyz, cell_xyz, NULL, g);
  for (int64_t p = 0; p < P; ++p) {
    v3 Pt = v3load(pt_xyz, p);
    int ix = bh_cell(&h, Pt.x), iy = bh_cell(&h, Pt.y), iz = bh_cell(&h, Pt.z);
    double best = INFINITY; int32_t bi = -1;
    int r, done = 0;
    for (r = 0; r <= 24 && !done; ++r) {
      for (int a = ix - r; a <= ix + r; ++a) for (int b = iy - r; b <= iy + r; ++b) for (int c = iz - r; c <= iz + r; ++c) {
        int cheb = abs(a - ix); if (abs(b - iy) > cheb) cheb = abs(b - iy); if (abs(c - iz) > cheb) cheb = abs(c - iz);
        if (cheb != r) continue;
        if (a < 0 || b < 0 || c < 0 || a >= h.G || b >= h.G || c >= h.G) continue;
        uint32_t k = bh_hash(&h, a, b, c);
        for (int64_t q = h.start[k]; q < h.start[k + 1]; ++q) {
          int32_t s = h.items[q];
          double d = dist2(Pt, v3load(cell_xyz, s));
          if (d < best || (d == best && s < bi)) { best = d; bi = s; }
        }
      }
      /* every unscanned site lies outside the cube of half-width r cells around P's cell: dist > r*g */
      double rg = r * h.g;
      if (bi >= 0 && best < rg * rg) done = 1;
    }
    if (!done) { /* far from the mesh: exhaustive */
      best = INFINITY; bi = -1;
      for (int64_t c = 0; c < nCells; ++c) {
        double d = dist2(Pt, v3load(cell_xyz, c));
        if (d < best) { best = d; bi = (int32_t)c; }
      }
    }
    idx[p] = bi;
  }
  bh_free(&h);
}

/* ------------------------------------------------------------------------------------------
 * A5. First-order conservative.  w_ij = Area(src_i ^ dst_j)/Area(dst_j); great-circle sides.
 * src polygon = Voronoi cell from vertex coords (verticesOnCell order); dst polygon = the 4
 * CORNER-stagger points around centre (i,j).
 * ---------------------------------------------------------------------------------------- */
#define ORC_MAXPOLY 32
/* signed spherical-triangle area (Van Oosterom & Strackee), difference form for conditioning */
static inline double sph_tri_area(v3 a, v3 b, v3 c) {
  double num = det3_from(a, b, c);
  double den = 1.0 + v3dot(a, b) + v3dot(b, c) + v3dot(c, a);
  return 2.0 * atan2(num, den);
}
static double sph_poly_area(int n, const v3 *p) {
  double s = 0;
  for (int i = 1; i + 1 < n; ++i) s += sph_tri_area(p[0], p[i], p[i + 1]);
  return s;
}
/* clip polygon (CCW seen from outside) against half-space n.X >= 0 */
static int clip_halfspace(int n, const v3 *in, v3 nrm, v3 *out) {
  int m = 0;
  double scale = sqrt(v3dot(nrm, nrm));
  double eps = 1e-15 * scale;
  for (int i = 0; i < n; ++i) {
    v3 X1 = in[i], X2 = in[(i + 1) % n];
    double d1 = v3dot(nrm, X1), d2 = v3dot(nrm, X2);
    int in1 = d1 >= -eps, in2 = d2 >= -eps;
    if (in1) out[m++] = X1;
    if (in1 != in2) {
      /* great-circle edge X1-X2 crosses the plane: X = X1*d2 - X2*d1 (normalised, on the arc) */
      v3 X = v3sub(v3scale(X1, d2), v3scale(X2, d1));
      double s = (d2 - d1) > 0 ? 1.0 : -1.0;
      X = v3scale(X, s);
      double nn = sqrt(v3dot(X, X));
      if (nn > 0) out[m++] = v3scale(X, 1.0 / nn);
    }
    if (m >= ORC_MAXPOLY - 1) break;
  }
  return m;
}
static double clip_area(int ns, const v3 *src, const v3 *quad) {
  v3 bufA[ORC_MAXPOLY], bufB[ORC_MAXPOLY];
  int n = ns;
  memcpy(bufA, src, sizeof(v3) * (size_t)ns);
  v3 *cur = bufA, *nxt = bufB;
  for (int e = 0; e < 4 && n >= 3; ++e) {
    /* collapsed side (both CORNER points of a lat-lon cell at a pole): bounds nothing, direction is noise */
    v3 side = v3sub(quad[(e + 1) & 3], quad[e]);
    if (v3dot(side, side) < 1e-24) continue;
    /* normal of the great circle through the side, in difference form: a x (b - a) = a x b, but the rounding error of the
     * direct product (1e-16 absolute on a vector of length |b - a|) would shift the plane by 1e-16 / |b - a| radians --
     * 4e-10 of a 3-km cell (tests/test_weight_goldens.py) */
    v3 nrm = v3cross(quad[e], side);
    n = clip_halfspace(n, cur, nrm, nxt);
    v3 *t = cur; cur = nxt; nxt = t;
  }
  if (n < 3) return 0.0;
  double a = sph_poly_area(n, cur);
  return a > 0 ? a : 0.0;
}

/* Two-call protocol: rowptr[Pdst+1] always filled; col/val filled when non-NULL (capacity cap).
 * corner_xyz is [(ny+1)][(nx+1)][3]; dst cell (i,j) uses corners (i,j),(i+1,j),(i+1,j+1),(i,j+1).
 * Row entries sorted by source cell id.  Returns total nnz.  */
int64_t orc_conserve(int64_t nCells, int64_t nVertices, int maxEdges, const int32_t *voc,
                     const double *vert_xyz, int nx, int ny, const double *corner_xyz,
                     int64_t *rowptr, int32_t *col, double *val, int64_t cap) {
  (void)nVertices;
  double *lo = (double *)malloc(sizeof(double) * 3 * (size_t)nCells), *hi = (double *)malloc(sizeof(double) * 3 * (size_t)nCells);
  uint8_t *valid = (uint8_t *)calloc((size_t)nCells, 1);
  double *diam = (double *)malloc(sizeof(double) * (size_t)(nCells + 1)); int64_t nd = 0;
  int8_t *flip = (int8_t *)calloc((size_t)nCells, 1);
  for (int64_t c = 0; c < nCells; ++c) {
    v3 poly[ORC_MAXPOLY]; int n = 0;
    for (int j = 0; j < maxEdges && n < ORC_MAXPOLY; ++j) { int32_t v = voc[c * maxEdges + j]; if (v > 0) poly[n++] = v3load(vert_xyz, v - 1); }
    if (n < 3) continue;
    double a = sph_poly_area(n, poly);
    if (a == 0) continue;
    flip[c] = a < 0; valid[c] = 1;
    double l[3] = {2, 2, 2}, u[3] = {-2, -2, -2}, e2 = 0;
    for (int i = 0; i < n; ++i) {
      l[0] = fmin(l[0], poly[i].x); l[1] = fmin(l[1], poly[i].y); l[2] = fmin(l[2], poly[i].z);
      u[0] = fmax(u[0], poly[i].x); u[1] = fmax(u[1], poly[i].y); u[2] = fmax(u[2], poly[i].z);
      v3 d = v3sub(poly[i], poly[0]); e2 = fmax(e2, v3dot(d, d));
    }
    double pad = 0.5 * e2 * 4 + 1e-9; /* diameter <= 2*max distance from vertex 0 */
    diam[nd++] = sqrt(e2);
    for (int k = 0; k < 3; ++k) { lo[3 * c + k] = l[k] - pad; hi[3 * c + k] = u[k] + pad; }
  }
  double g = 0.05;
  if (nd > 0) { qsort(diam, (size_t)nd, sizeof(double), cmp_double); g = 1.5 * diam[nd / 2]; }
  boxhash h; bh_build(&h, nCells, lo, hi, valid, g);
  int64_t nnz = 0;
  const int64_t cand_cap = nCells + 1 > 65536 ? nCells + 1 : 65536;
  int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)cand_cap);
  int nxc = nx + 1;
  for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
    int64_t p = (int64_t)j * nx + i;
    rowptr[p] = nnz;
    v3 q[4] = {v3load(corner_xyz, (int64_t)j * nxc + i), v3load(corner_xyz, (int64_t)j * nxc + i + 1),
               v3load(corner_xyz, (int64_t)(j + 1) * nxc + i + 1), v3load(corner_xyz, (int64_t)(j + 1) * nxc + i)};
    double aq = sph_poly_area(4, q);
    if (aq < 0) { v3 t = q[1]; q[1] = q[3]; q[3] = t; aq = -aq; }
    if (!(aq > 0)) continue;
    double l[3] = {2, 2, 2}, u[3] = {-2, -2, -2}, e2 = 0;
    for (int k = 0; k < 4; ++k) {
      l[0] = fmin(l[0], q[k].x); l[1] = fmin(l[1], q[k].y); l[2] = fmin(l[2], q[k].z);
      u[0] = fmax(u[0], q[k].x); u[1] = fmax(u[1], q[k].y); u[2] = fmax(u[2], q[k].z);
      v3 d = v3sub(q[k], q[0]); e2 = fmax(e2, v3dot(d, d));
    }
    double pad = 2.0 * e2 + 1e-9;
    int a0 = bh_cell(&h, l[0] - pad), a1 = bh_cell(&h, u[0] + pad), b0 = bh_cell(&h, l[1] - pad), b1 = bh_cell(&h, u[1] + pad),
        c0 = bh_cell(&h, l[2] - pad), c1 = bh_cell(&h, u[2] + pad);
    int64_t nc = 0;
    int overflow = (int64_t)(a1 - a0 + 1) * (b1 - b0 + 1) * (c1 - c0 + 1) > 4096;  /* huge destination cell: scan every source cell */
    for (int a = a0; a <= a1 && !overflow; ++a) for (int b = b0; b <= b1 && !overflow; ++b) for (int c = c0; c <= c1 && !overflow; ++c) {
      uint32_t k = bh_hash(&h, a, b, c);
      for (int64_t t = h.start[k]; t < h.start[k + 1]; ++t) {
        if (nc >= cand_cap) { overflow = 1; break; }
        cand[nc++] = h.items[t];
      }
    }
    for (int64_t t = 0; t < h.nbig && !overflow; ++t) {
      if (nc >= cand_cap) { overflow = 1; break; }
      cand[nc++] = h.big[t];
    }
    if (overflow) {  /* never truncate the candidate set */
      nc = 0;
      for (int64_t c = 0; c < nCells; ++c) if (valid[c]) cand[nc++] = (int32_t)c;
    } else {  /* sort + unique */
      for (int64_t a = 1; a < nc; ++a) { int32_t key = cand[a]; int64_t b = a - 1; while (b >= 0 && cand[b] > key) { cand[b + 1] = cand[b]; --b; } cand[b + 1] = key; }
    }
    int32_t last = -1;
    for (int64_t a = 0; a < nc; ++a) {
      int32_t c = cand[a];
      if (c == last) continue;
      last = c;
      if (hi[3 * c] < l[0] - pad || lo[3 * c] > u[0] + pad || hi[3 * c + 1] < l[1] - pad || lo[3 * c + 1] > u[1] + pad || hi[3 * c + 2] < l[2] - pad || lo[3 * c + 2] > u[2] + pad) continue;
      v3 poly[ORC_MAXPOLY]; int n = 0;
      for (int jj = 0; jj < maxEdges && n < ORC_MAXPOLY; ++jj) { int32_t v = voc[(int64_t)c * maxEdges + jj]; if (v > 0) poly[n++] = v3load(vert_xyz, v - 1); }
      if (flip[c]) for (int a2 = 0; a2 < n / 2; ++a2) { v3 t = poly[a2]; poly[a2] = poly[n - 1 - a2]; poly[n - 1 - a2] = t; }
      double ar = clip_area(n, poly, q);
      if (ar > 1e-14 * aq) {
        if (col && nnz < cap) { col[nnz] = c; val[nnz] = ar / aq; }
        nnz++;
      }
    }
  }
  rowptr[(int64_t)nx * ny] = nnz;
  free(cand); bh_free(&h); free(lo); free(hi); free(valid); free(diam); free(flip);
  return nnz;
}

/* ------------------------------------------------------------------------------------------
 * A4. Grid->Grid bilinear (CENTER -> EDGE1/EDGE2).  Source cells = quads of 4 neighbouring
 * CENTER points; weights from X(xi,eta) = t*P solved by Newton in 3-D.  stagger: 1 = EDGE1
 * ((nx+1) x ny points, U at (i-1/2, j)), 2 = EDGE2 (nx x (ny+1), V at (i, j-1/2)).
 * idx[4*p+k] = flat centre index (j*nx+i) or -1; lowest quad id among containing quads.
 * ---------------------------------------------------------------------------------------- */
static int quad_solve(v3 P, v3 A, v3 B, v3 C, v3 D, double *xi, double *eta) {
  double s = 0.5, t = 0.5, lam;
  /* X(s,t) = A + s(B-A) + t(D-A) + st(A-B+C-D);  F = X - lam*P */
  v3 e1 = v3sub(B, A), e2 = v3sub(D, A), e3 = v3add(v3sub(A, B), v3sub(C, D));
  lam = 1.0;
  for (int it = 0; it < 50; ++it) {
    v3 X = v3add(v3add(A, v3scale(e1, s)), v3add(v3scale(e2, t), v3scale(e3, s * t)));
    v3 F = v3sub(X, v3scale(P, lam));
    v3 Js = v3add(e1, v3scale(e3, t)), Jt = v3add(e2, v3scale(e3, s)), Jl = v3scale(P, -1.0);
    /* solve [Js Jt Jl] d = -F by Cramer */
    double det = v3dot(Js, v3cross(Jt, Jl));
    if (det == 0) return 0;
    v3 mF = v3scale(F, -1.0);
    double ds = v3dot(mF, v3cross(Jt, Jl)) / det;
    double dt = v3dot(Js, v3cross(mF, Jl)) / det;
    double dl = v3dot(Js, v3cross(Jt, mF)) / det;
    s += ds; t += dt; lam += dl;
    /* quadratic convergence: a step below 1e-9 leaves ~1e-18, far under the rounding noise of the residual (2e-13 in xi, eta
     * for 3 km cells); 1e-15 was below that noise and never met */
    if (fabs(ds) < 1e-9 && fabs(dt) < 1e-9) break;
  }
  *xi = s; *eta = t;
  return lam > 0;
}

/* flags = 0: ESMF_GridCreateNoPeriDim.  flags & 1: ESMF_GridCreate1PeriDim(periodicDim=1, poleDim=2, MONOPOLE)
 * (model_grid.F90:685-694): the CENTER columns wrap, and each j end is closed by a pole node at lat -/+90 whose
 * value is the mean of the first / last CENTER row; a destination point in the cap triangle (pole, A, B) gets
 * planar barycentric weights (A2) on A, B and the pole.  flags & 2 / & 4: no south / north cap (row block of a
 * periodic grid).  pole_w [2][nxd] (south candidate row j = 0, north candidate row j = nyd-1) receives the pole
 * weight, pole_src0 the first source of the averaged row; both may be NULL when flags == 0. */
void orc_grid_bilinear_p(int nx, int ny, int flags, const double *centre_xyz, int stagger, const double *dst_xyz,
                         int32_t *idx, double *w, int32_t *pole_src0, double *pole_w) {
  int nxd = stagger == 1 ? nx + 1 : nx, nyd = stagger == 2 ? ny + 1 : ny;
  int per = flags & 1;
  if (per) for (int q = 0; q < 2 * nxd; ++q) { pole_src0[q] = 0; pole_w[q] = 0.0; }
  for (int j = 0; j < nyd; ++j) for (int i = 0; i < nxd; ++i) {
    int64_t p = (int64_t)j * nxd + i;
    v3 P = v3load(dst_xyz, p);
    int ca[2], cb[2], nca, ncb;
    if (stagger == 1) { ca[0] = i - 1; nca = 1; cb[0] = j - 1; cb[1] = j; ncb = 2; }
    else { ca[0] = i - 1; ca[1] = i; nca = 2; cb[0] = j - 1; ncb = 1; }
    int found = 0;
    for (int bb = 0; bb < ncb && !found; ++bb) for (int aa = 0; aa < nca && !found; ++aa) {
      int a = ca[aa], b = cb[bb];
      if (per) a = (a + nx) % nx;
      if (a < 0 || b < 0 || b + 1 >= ny || (!per && a + 1 >= nx)) continue;
      int a1 = a + 1 == nx ? 0 : a + 1;
      int64_t iA = (int64_t)b * nx + a, iB = (int64_t)b * nx + a1, iC = iB + nx, iD = iA + nx;
      double xi, eta;
      if (!quad_solve(P, v3load(centre_xyz, iA), v3load(centre_xyz, iB), v3load(centre_xyz, iC), v3load(centre_xyz, iD), &xi, &eta)) continue;
      if (xi < -g_orc_grid_tol || xi > 1 + g_orc_grid_tol || eta < -g_orc_grid_tol || eta > 1 + g_orc_grid_tol) continue;
      idx[4 * p] = (int32_t)iA; idx[4 * p + 1] = (int32_t)iB; idx[4 * p + 2] = (int32_t)iC; idx[4 * p + 3] = (int32_t)iD;
      w[4 * p] = (1 - xi) * (1 - eta); w[4 * p + 1] = xi * (1 - eta); w[4 * p + 2] = xi * eta; w[4 * p + 3] = (1 - xi) * eta;
      found = 1;
    }
    if (!found) for (int k = 0; k < 4; ++k) { idx[4 * p + k] = -1; w[4 * p + k] = 0; }
    if (per && !found)
      for (int bb = 0; bb < ncb && !found; ++bb) for (int aa = 0; aa < nca && !found; ++aa) {
        int b = cb[bb];
        int south = b == -1 && !(flags & 2), north = b == ny - 1 && !(flags & 4);
        if (!south && !north) continue;
        int a = (ca[aa] + nx) % nx, a1 = a + 1 == nx ? 0 : a + 1;
        int64_t row0 = south ? 0 : (int64_t)(ny - 1) * nx;
        v3 A = v3load(centre_xyz, row0 + a), B = v3load(centre_xyz, row0 + a1);
        v3 N = {0, 0, 1}, S = {0, 0, -1};
        double t[3];
        /* counter-clockwise seen from outside: (A, B, N) / (B, A, S) */
        int in = north ? tri_weights(P, A, B, N, ORC_TOL, t) : tri_weights(P, B, A, S, ORC_TOL, t);
        if (!in) continue;
        idx[4 * p] = (int32_t)(row0 + a); idx[4 * p + 1] = (int32_t)(row0 + a1);
        w[4 * p] = north ? t[0] : t[1]; w[4 * p + 1] = north ? t[1] : t[0];
        int64_t q = (j == 0 ? 0 : nxd) + i;
        pole_src0[q] = (int32_t)row0;
        pole_w[q] = t[2];
        found = 1;
      }
  }
}
void orc_grid_bilinear(int nx, int ny, const double *centre_xyz, int stagger, const double *dst_xyz,
                       int32_t *idx, double *w) {
  orc_grid_bilinear_p(nx, ny, 0, centre_xyz, stagger, dst_xyz, idx, w, 0, 0);
}

/* ------------------------------------------------------------------------------------------
 * A7. Application.  dst(p,k) = sum_j w_pj * src(c_pj, k), accumulated in float64 in index order;
 * unmapped => 0.0.  src is cell-fastest [nlev][nsrc] (input_data.F90:653-655) or level-fastest
 * [nsrc][nlev] (file layout, input_data.F90:630,645); dst is [nlev][P].
 * ---------------------------------------------------------------------------------------- */
void orc_apply_fixed(int nnz_per_row, int64_t P, const int32_t *idx, const double *w, int64_t nsrc, int nlev,
                     int lev_fast, const double *src, double *dst) {
  for (int k = 0; k < nlev; ++k)
    for (int64_t p = 0; p < P; ++p) {
      double acc = 0.0;
      int mapped = 0;
      for (int j = 0; j < nnz_per_row; ++j) {
        int32_t c = idx[nnz_per_row * p + j];
        if (c < 0) continue;
        mapped = 1;
        double s = lev_fast ? src[(int64_t)c * nlev + k] : src[(int64_t)k * nsrc + c];
        acc += w[nnz_per_row * p + j] * s;
      }
      dst[(int64_t)k * P + p] = mapped ? acc : 0.0;
    }
}
/* nearest: pure copy (bit exact) */
void orc_apply_nearest(int64_t P, const int32_t *idx, int64_t nsrc, int nlev, int lev_fast, const double *src, double *dst) {
  for (int k = 0; k < nlev; ++k)
    for (int64_t p = 0; p < P; ++p) {
      int32_t c = idx[p];
      dst[(int64_t)k * P + p] = c < 0 ? 0.0 : (lev_fast ? src[(int64_t)c * nlev + k] : src[(int64_t)k * nsrc + c]);
    }
}
void orc_apply_csr(int64_t P, const int64_t *rowptr, const int32_t *col, const double *val, int64_t nsrc, int nlev,
                   int lev_fast, const double *src, double *dst) {
  for (int k = 0; k < nlev; ++k)
    for (int64_t p = 0; p < P; ++p) {
      double acc = 0.0;
      for (int64_t q = rowptr[p]; q < rowptr[p + 1]; ++q) {
        int32_t c = col[q];
        acc += val[q] * (lev_fast ? src[(int64_t)c * nlev + k] : src[(int64_t)k * nsrc + c]);
      }
      dst[(int64_t)k * P + p] = acc;
    }
}

/* Threaded variant of the 3-point apply used only as bench.py's cpu_baseline ("port"): same arithmetic as
 * orc_apply_fixed(3, ...).  Blocked by TARGET TILE (round 4): a thread takes 2048 consecutive target points and runs all
 * levels over them, so the block's indices and weights (72 KB) stay in its L2 for the 55 levels and the source cells a
 * block references (a stretch of a few thousand neighbouring ids per level) are read as near-contiguous lines.  The
 * round-1 form looped levels outermost and re-read the 36 B of indices + weights per point for every level: 3.8 GB of
 * the 5.9 GB it moved per configuration-4 field, ~1 GB/s per core. */
void orc_apply3_mt(int64_t P, const int32_t *idx, const double *w, int64_t nsrc, int nlev, const double *src, double *dst) {
  const int64_t B = 2048, nb = (P + B - 1) / B;
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t b = 0; b < nb; ++b) {
    const int64_t pb = b * B, pe = pb + B < P ? pb + B : P;
    for (int k = 0; k < nlev; ++k) {
      const double *s = src + (int64_t)k * nsrc;
      double *d = dst + (int64_t)k * P;
      for (int64_t p = pb; p < pe; ++p) {
        int32_t c0 = idx[3 * p], c1 = idx[3 * p + 1], c2 = idx[3 * p + 2];
        d[p] = c0 < 0 ? 0.0 : ((w[3 * p] * s[c0] + w[3 * p + 1] * s[c1]) + w[3 * p + 2] * s[c2]);
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * rotate_winds_cgrid (interp.F90:737-748): in place, per (i,j), all levels.
 *   tana = sina/cosa; u' = (u + v*tana)/(cosa + sina*tana); v' = (v - u'*sina)/cosa
 * ---------------------------------------------------------------------------------------- */
void orc_rotate_winds(int64_t npts, int nlev, const double *cosa, const double *sina, double *u, double *v) {
  for (int k = 0; k < nlev; ++k)
    for (int64_t p = 0; p < npts; ++p) {
      double tana = sina[p] / cosa[p];
      int64_t q = (int64_t)k * npts + p;
      double un = (u[q] + v[q] * tana) / (cosa[p] + sina[p] * tana);
      double vn = (v[q] - un * sina[p]) / cosa[p];
      u[q] = un; v[q] = vn;
    }
}

/* ------------------------------------------------------------------------------------------
 * Target-grid coordinates ("params" path): module_map_utils.F90 map_set/set_lc/lc_cone/ijll_lc/
 * llij_lc/ijll_latlon + llxy_module.F90 xytoll + model_grid.F90 get_lat_lon_fields/get_rotang.
 * All reals are float64 (CMakeLists.txt:80-82).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int code;            /* 0 = LATLON, 1 = LC, 2 = PS, 3 = MERC (misc_definitions_module.F90:38-42) */
  double lat1, lon1, knowni, knownj, dx, stdlon, truelat1, truelat2, hemi, cone, polei, polej, rsw, rebydx;
  double latinc, loninc; int nxmin, nxmax;
  double dlon;         /* PROJ_MERC (set_merc) */
} orc_proj;

static double wrap180(double x) {
  int it = 0;
  while (fabs(x) > 180. && it < 10) { if (x < -180.) x += 360.; if (x > 180.) x -= 360.; ++it; }
  return x;
}

/* module_map_utils.F90:1124-1157 */
static double lc_cone(double truelat1, double truelat2) {
  if (fabs(truelat1 - truelat2) > 0.1) {
    double cone = log10(cos(truelat1 * ORC_RAD_PER_DEG)) - log10(cos(truelat2 * ORC_RAD_PER_DEG));
    cone = cone / (log10(tan((45.0 - fabs(truelat1) / 2.0) * ORC_RAD_PER_DEG)) - log10(tan((45.0 - fabs(truelat2) / 2.0) * ORC_RAD_PER_DEG)));
    return cone;
  }
  return sin(fabs(truelat1) * ORC_RAD_PER_DEG);
}

/* map_set(PROJ_LC,...) module_map_utils.F90:243-567 + set_lc :1083-1121 */
void orc_map_set_lc(orc_proj *p, double truelat1, double truelat2, double stdlon, double lat1, double lon1,
                    double knowni, double knownj, double dx) {
  memset(p, 0, sizeof(*p));
  p->code = 1;
  p->lat1 = lat1; p->lon1 = wrap180(lon1); p->knowni = knowni; p->knownj = knownj; p->dx = dx;
  p->stdlon = wrap180(stdlon); p->truelat1 = truelat1; p->truelat2 = truelat2;
  p->hemi = truelat1 < 0. ? -1.0 : 1.0;
  p->rebydx = ORC_EARTH_RADIUS_M / dx;
  if (fabs(p->truelat2) > 90.) p->truelat2 = p->truelat1;
  p->cone = lc_cone(p->truelat1, p->truelat2);
  double deltalon1 = p->lon1 - p->stdlon;
  if (deltalon1 > 180.) deltalon1 -= 360.;
  if (deltalon1 < -180.) deltalon1 += 360.;
  double tl1r = p->truelat1 * ORC_RAD_PER_DEG, ctl1r = cos(tl1r);
  p->rsw = p->rebydx * ctl1r / p->cone *
           pow(tan((90. * p->hemi - p->lat1) * ORC_RAD_PER_DEG / 2.) / tan((90. * p->hemi - p->truelat1) * ORC_RAD_PER_DEG / 2.), p->cone);
  double arg = p->cone * (deltalon1 * ORC_RAD_PER_DEG);
  p->polei = p->hemi * p->knowni - p->hemi * p->rsw * sin(arg);
  p->polej = p->hemi * p->knownj + p->rsw * cos(arg);
}

/* map_set(PROJ_LATLON,...) as called from llxy_module.F90:60-69 */
void orc_map_set_latlon(orc_proj *p, double lat1, double lon1, double knowni, double knownj, double latinc, double loninc) {
  memset(p, 0, sizeof(*p));
  p->code = 0; p->lat1 = lat1; p->lon1 = wrap180(lon1); p->knowni = knowni; p->knownj = knownj;
  p->latinc = latinc; p->loninc = loninc; p->nxmin = 1; p->nxmax = (int)lround(360.0 / loninc);
}

/* map_set(PROJ_PS,...) as called from llxy_module.F90:123-132 + set_ps (module_map_utils.F90:682-715) */
void orc_map_set_ps(orc_proj *p, double truelat1, double stdlon, double lat1, double lon1, double knowni, double knownj, double dx) {
  memset(p, 0, sizeof(*p));
  p->code = 2;
  p->lat1 = lat1; p->lon1 = wrap180(lon1); p->knowni = knowni; p->knownj = knownj; p->dx = dx;
  p->stdlon = wrap180(stdlon); p->truelat1 = truelat1;
  p->hemi = truelat1 < 0. ? -1.0 : 1.0;
  p->rebydx = ORC_EARTH_RADIUS_M / dx;
  double reflon = p->stdlon + 90.;
  double scale_top = 1. + p->hemi * sin(p->truelat1 * ORC_RAD_PER_DEG);
  double ala1 = p->lat1 * ORC_RAD_PER_DEG;
  p->rsw = p->rebydx * cos(ala1) * scale_top / (1. + p->hemi * sin(ala1));
  double alo1 = (p->lon1 - reflon) * ORC_RAD_PER_DEG;
  p->polei = p->knowni - p->rsw * cos(alo1);
  p->polej = p->knownj - p->hemi * p->rsw * sin(alo1);
}

/* map_set(PROJ_MERC,...) as called from llxy_module.F90:71-79 + set_merc (module_map_utils.F90:1293-1317) */
void orc_map_set_merc(orc_proj *p, double truelat1, double lat1, double lon1, double knowni, double knownj, double dx) {
  memset(p, 0, sizeof(*p));
  p->code = 3;
  p->lat1 = lat1; p->lon1 = wrap180(lon1); p->knowni = knowni; p->knownj = knownj; p->dx = dx; p->truelat1 = truelat1;
  p->hemi = truelat1 < 0. ? -1.0 : 1.0;
  p->rebydx = ORC_EARTH_RADIUS_M / dx;
  double clain = cos(ORC_RAD_PER_DEG * p->truelat1);
  p->dlon = dx / (ORC_EARTH_RADIUS_M * clain);
  p->rsw = 0.;
  if (p->lat1 != 0.) p->rsw = log(tan(0.5 * ((p->lat1 + 90.) * ORC_RAD_PER_DEG))) / p->dlon;
}

/* module_map_utils.F90:763-822 */
static void ijll_ps(const orc_proj *p, double i, double j, double *lat, double *lon) {
  double reflon = p->stdlon + 90.;
  double scale_top = 1. + p->hemi * sin(p->truelat1 * ORC_RAD_PER_DEG);
  double xx = i - p->polei, yy = (j - p->polej) * p->hemi;
  double r2 = xx * xx + yy * yy;
  if (r2 == 0.) { *lat = p->hemi * 90.; *lon = reflon; }
  else {
    double gi2 = pow(p->rebydx * scale_top, 2.);
    *lat = ORC_DEG_PER_RAD * p->hemi * asin((gi2 - r2) / (gi2 + r2));
    double c = xx / sqrt(r2);
    if (c < -1.) c = -1.;
    if (c > 1.) c = 1.;
    double arccos = acos(c);
    *lon = yy > 0 ? reflon + ORC_DEG_PER_RAD * arccos : reflon - ORC_DEG_PER_RAD * arccos;
  }
  if (*lon > 180.) *lon -= 360.;
  if (*lon < -180.) *lon += 360.;
}
/* module_map_utils.F90:718-760 */
static void llij_ps(const orc_proj *p, double lat, double lon, double *i, double *j) {
  double reflon = p->stdlon + 90.;
  double scale_top = 1. + p->hemi * sin(p->truelat1 * ORC_RAD_PER_DEG);
  double ala = lat * ORC_RAD_PER_DEG;
  double rm = p->rebydx * cos(ala) * scale_top / (1. + p->hemi * sin(ala));
  double alo = (lon - reflon) * ORC_RAD_PER_DEG;
  *i = p->polei + rm * cos(alo);
  *j = p->polej + p->hemi * rm * sin(alo);
}
/* module_map_utils.F90:1344-1362 */
static void ijll_merc(const orc_proj *p, double i, double j, double *lat, double *lon) {
  *lat = 2.0 * atan(exp(p->dlon * (p->rsw + j - p->knownj))) * ORC_DEG_PER_RAD - 90.;
  *lon = (i - p->knowni) * p->dlon * ORC_DEG_PER_RAD + p->lon1;
  if (*lon > 180.) *lon -= 360.;
  if (*lon < -180.) *lon += 360.;
}
/* module_map_utils.F90:1320-1341 */
static void llij_merc(const orc_proj *p, double lat, double lon, double *i, double *j) {
  double deltalon = lon - p->lon1;
  if (deltalon < -180.) deltalon += 360.;
  if (deltalon > 180.) deltalon -= 360.;
  *i = p->knowni + (deltalon / (p->dlon * ORC_DEG_PER_RAD));
  *j = p->knownj + log(tan(0.5 * ((lat + 90.) * ORC_RAD_PER_DEG))) / p->dlon - p->rsw;
}

/* module_map_utils.F90:1160-1233 */
static void ijll_lc(const orc_proj *p, double i, double j, double *lat, double *lon) {
  double chi1 = (90. - p->hemi * p->truelat1) * ORC_RAD_PER_DEG;
  double chi2 = (90. - p->hemi * p->truelat2) * ORC_RAD_PER_DEG;
  double inew = p->hemi * i, jnew = p->hemi * j;
  double xx = inew - p->polei, yy = p->polej - jnew;
  double r2 = xx * xx + yy * yy, r = sqrt(r2) / p->rebydx;
  if (r2 == 0.) { *lat = p->hemi * 90.; *lon = p->stdlon; }
  else {
    double lo = p->stdlon + ORC_DEG_PER_RAD * atan2(p->hemi * xx, yy) / p->cone;
    lo = fmod(lo + 360., 360.);
    double chi;
    if (chi1 == chi2) chi = 2.0 * atan(pow(r / tan(chi1), 1. / p->cone) * tan(chi1 * 0.5));
    else chi = 2.0 * atan(pow(r * p->cone / sin(chi1), 1. / p->cone) * tan(chi1 * 0.5));
    *lat = (90.0 - chi * ORC_DEG_PER_RAD) * p->hemi;
    *lon = lo;
  }
  if (*lon > 180.) *lon -= 360.;
  if (*lon < -180.) *lon += 360.;
}
/* module_map_utils.F90:1236-1290 */
static void llij_lc(const orc_proj *p, double lat, double lon, double *i, double *j) {
  double deltalon = lon - p->stdlon;
  if (deltalon > 180.) deltalon -= 360.;
  if (deltalon < -180.) deltalon += 360.;
  double tl1r = p->truelat1 * ORC_RAD_PER_DEG, ctl1r = cos(tl1r);
  double rm = p->rebydx * ctl1r / p->cone *
              pow(tan((90. * p->hemi - lat) * ORC_RAD_PER_DEG / 2.) / tan((90. * p->hemi - p->truelat1) * ORC_RAD_PER_DEG / 2.), p->cone);
  double arg = p->cone * (deltalon * ORC_RAD_PER_DEG);
  *i = p->hemi * (p->polei + p->hemi * rm * sin(arg));
  *j = p->hemi * (p->polej - rm * cos(arg));
}
/* module_map_utils.F90:1398-1428 */
static void ijll_latlon(const orc_proj *p, double i, double j, double *lat, double *lon) {
  double i_work = i;
  if (i < (double)p->nxmin - 0.5) i_work = i + (double)(p->nxmax - p->nxmin + 1);
  if (i >= (double)p->nxmax + 0.5) i_work = i - (double)(p->nxmax - p->nxmin + 1);
  i_work -= p->knowni;
  double j_work = j - p->knownj;
  *lat = p->lat1 + j_work * p->latinc;
  *lon = p->lon1 + i_work * p->loninc;
}

void orc_ij_to_latlon(const orc_proj *p, double i, double j, double *lat, double *lon) {
  if (p->code == 1) ijll_lc(p, i, j, lat, lon);
  else if (p->code == 2) ijll_ps(p, i, j, lat, lon);
  else if (p->code == 3) ijll_merc(p, i, j, lat, lon);
  else ijll_latlon(p, i, j, lat, lon);
}
/* latlon_to_ij (module_map_utils.F90:570-626) for the projected grids */
void orc_latlon_to_ij(const orc_proj *p, double lat, double lon, double *i, double *j) {
  if (p->code == 2) llij_ps(p, lat, lon, i, j);
  else if (p->code == 3) llij_merc(p, lat, lon, i, j);
  else llij_lc(p, lat, lon, i, j);
}
/* get_map_factor (model_grid.F90:2229-2365) at one latitude: LC (one / two true latitudes), PS, MERC */
double orc_map_factor(const orc_proj *p, double xlat) {
  if (p->code == 1) {
    double colat = ORC_RAD_PER_DEG * (90.0 - xlat);
    if (p->truelat1 != p->truelat2) {
      double colat1 = ORC_RAD_PER_DEG * (90.0 - p->truelat1), colat2 = ORC_RAD_PER_DEG * (90.0 - p->truelat2);
      double n = (log(sin(colat1)) - log(sin(colat2))) / (log(tan(colat1 / 2.0)) - log(tan(colat2 / 2.0)));
      return sin(colat2) / sin(colat) * pow(tan(colat / 2.0) / tan(colat2 / 2.0), n);
    }
    double colat0 = ORC_RAD_PER_DEG * (90.0 - p->truelat1);
    return sin(colat0) / sin(colat) * pow(tan(colat / 2.0) / tan(colat0 / 2.0), cos(colat0));
  }
  if (p->code == 2) return (1.0 + sin(ORC_RAD_PER_DEG * fabs(p->truelat1))) / (1.0 + sin(ORC_RAD_PER_DEG * copysign(1., p->truelat1) * xlat));
  if (p->code == 3) return sin(ORC_RAD_PER_DEG * (90.0 - p->truelat1)) / sin(ORC_RAD_PER_DEG * (90.0 - xlat));
  return 1.0;
}
void orc_latlon_to_ij_lc(const orc_proj *p, double lat, double lon, double *i, double *j) { llij_lc(p, lat, lon, i, j); }

/* xytoll (llxy_module.F90:166-216); stagger codes: M=1,U=2,V=3,CORNER=4 here */
void orc_xytoll(const orc_proj *p, double x, double y, int stagger, double *lat, double *lon) {
  double rx = x, ry = y;
  if (stagger == 2) rx = x - 0.5;
  else if (stagger == 3) ry = y - 0.5;
  else if (stagger == 4) { rx = x - 0.5; ry = y - 0.5; }
  orc_ij_to_latlon(p, rx, ry, lat, lon);
}

/* get_lat_lon_fields (model_grid.F90:2188-2219): point (i,j), 1-based, -> xytoll(i, j, stagger);
 * output arrays [nj][ni] (C order, i fastest), i = 1..ni, j = 1..nj. */
void orc_lat_lon_fields(const orc_proj *p, int ni, int nj, int stagger, double *lat, double *lon) {
  for (int j = 1; j <= nj; ++j)
    for (int i = 1; i <= ni; ++i) {
      double x = (double)(i - 0.5) / 1.0 + 0.5, y = (double)(j - 0.5) / 1.0 + 0.5;
      orc_xytoll(p, x, y, stagger, &lat[(int64_t)(j - 1) * ni + (i - 1)], &lon[(int64_t)(j - 1) * ni + (i - 1)]);
    }
}

/* get_rotang (model_grid.F90:2450-2507) on an [nj][ni] grid */
void orc_get_rotang(int ni, int nj, const double *xlat, const double *xlon, double *cosa, double *sina) {
#define AT(a, i, j) a[(int64_t)(j) * ni + (i)]
  for (int i = 0; i < ni; ++i) {
    for (int j = 0; j < nj; ++j) {
      int jm = j - 1, jp = j + 1;
      if (j == 0) jm = 0;
      if (j == nj - 1) jp = nj - 1;
      double d_lon = AT(xlon, i, jp) - AT(xlon, i, jm);
      if (d_lon > 180.) d_lon -= 360.; else if (d_lon < -180.) d_lon += 360.;
      double alpha = atan2(-cos(AT(xlat, i, j) * ORC_RAD_PER_DEG) * (d_lon * ORC_RAD_PER_DEG),
                           (AT(xlat, i, jp) - AT(xlat, i, jm)) * ORC_RAD_PER_DEG);
      AT(sina, i, j) = sin(alpha); AT(cosa, i, j) = cos(alpha);
    }
  }
#undef AT
}

/* para_range (model_grid.F90:2428-2441) */
void orc_para_range(int n1, int n2, int nprocs, int irank, int *ista, int *iend) {
  int iwork1 = (n2 - n1 + 1) / nprocs, iwork2 = (n2 - n1 + 1) % nprocs;
  *ista = irank * iwork1 + n1 + (irank < iwork2 ? irank : iwork2);
  *iend = *ista + iwork1 - 1;
  if (iwork2 > irank) *iend += 1;
}

int orc_sizeof_proj(void) { return (int)sizeof(orc_proj); }
