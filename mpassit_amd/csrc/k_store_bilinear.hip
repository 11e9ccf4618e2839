// K1 "build_tri_weights": bilinear RegridStore for cell-centred sources (ESMF_MESHLOC_ELEMENT).
//
// Replaces ESMF_Field[Bundle]RegridStore(regridmethod=BILINEAR) at interp.F90:123,207,226,241,259,
// 277,334 (mesh -> CENTER stagger).  Semantics: the source "cells" are the dual
// (Delaunay) triangles of the MPAS mesh; a target point P takes the 3 gnomonic-barycentric weights of
// the triangle that contains it; points in no triangle stay unmapped (-> 0.0 on Regrid).
//
// MI355X-native formulation: the structured target grid is treated as a framebuffer and the triangles
// are rasterised onto it.  One thread per triangle walks an AABB pyramid built over the target points
// (k_setup.hip), tests the points of the leaf blocks it overlaps and claims them with
// atomicMin(owner[p], triangle id) -- lowest id wins on shared edges, so the result is deterministic.
// A second pass (one thread per target point) recomputes the weights of the winning triangle and
// stores them SoA ([3][P]) for the coalesced apply kernel.  No sort, no hash, no fallback path:
// cost is O(T log P + P) and the traversal is exact for any mesh (no Delaunay assumption).
// No floating-point contraction in this translation unit (see k_store_conserve.hip): what it computes -- weights, coordinates --
// is a function of the source text, not of which product the compiler chooses to fuse; explicit fma() calls stay what they are.
#pragma clang fp contract(off)
#include "geom.h"
#include <algorithm>

#include "mpg_internal.h"

#define RASTER_STACK 64
#define RASTER_BIG_LEAVES 16   // 4 x 4-point leaves a lane rasterises by itself before it hands its triangle over

template <bool NORMAL>
__global__ __launch_bounds__(256) void k_tri_raster(int64_t nTri, const int32_t *__restrict__ tri, int64_t triStride,
                                                    const double *__restrict__ cx, const double *__restrict__ cy,
                                                    const double *__restrict__ cz, PyramidView pyr, int npx, int npy,
                                                    const double *__restrict__ px, const double *__restrict__ py,
                                                    const double *__restrict__ pz, int32_t *__restrict__ owner,
                                                    int32_t *__restrict__ overflow, int32_t *__restrict__ big, int32_t *__restrict__ nbig,
                                                    int big_cap, const float *__restrict__ sij, float di, float dj, float pad_coef, float pad_latlon, float e_max) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= nTri) return;
  int32_t ia = tri[t];
  if (ia < 0) return;
  int32_t ib = tri[triStride + t], ic = tri[2 * triStride + t];
  dv3 A = ld3(cx, cy, cz, ia), B = ld3(cx, cy, cz, ib), C = ld3(cx, cy, cz, ic);
  // O(1) candidates on a projection-built grid (round 4): the triangle's corners in the grid's own index space (sij: the inverse
  // projection of every source point, k_target_grid.hip; di / dj: offset of this stagger's point indices) bound the target points
  // it can hold -- a box of a few points around it, padded for the bend of its edges on the map and for the float32 indices.
  // Every point of the box gets the very test of the pyramid's leaves below, so the owners are the same; a triangle whose
  // corners have no usable index (near the projection's pole or cut, poleward of 85 degrees on a lat-lon grid), that spans more
  // than e_max index units (six degrees, sixteen units at most) or whose box holds more than 256 points takes the walk.
  if (sij) {
    const float ai = sij[2 * ia], aj = sij[2 * ia + 1], bi_ = sij[2 * ib], bj_ = sij[2 * ib + 1], ci_ = sij[2 * ic], cj_ = sij[2 * ic + 1];
    const float imin = fminf(ai, fminf(bi_, ci_)), imax = fmaxf(ai, fmaxf(bi_, ci_)), jmin = fminf(aj, fminf(bj_, cj_)), jmax = fmaxf(aj, fmaxf(bj_, cj_));
    const float E = fmaxf(imax - imin, jmax - jmin);
    const bool usable = ai == ai && aj == aj && bi_ == bi_ && bj_ == bj_ && ci_ == ci_ && cj_ == cj_;   // fminf / fmaxf skip a NaN corner
    if (!usable && big_cap > 0) {
      // no usable index on a grid that has one -- the last degrees before a pole, the far side of a Lambert projection: such a
      // triangle covers hundreds of thin cells or none at all; a wavefront takes it (k_tri_raster_big) unless it misses the grid
      const double *bx = pyr.box + 6 * pyr.off[pyr.nlev - 1];
      const double xl = fmin(A.x, fmin(B.x, C.x)), xh = fmax(A.x, fmax(B.x, C.x)), yl = fmin(A.y, fmin(B.y, C.y)), yh = fmax(A.y, fmax(B.y, C.y)),
                   zl = fmin(A.z, fmin(B.z, C.z)), zh = fmax(A.z, fmax(B.z, C.z));
      dv3 ab0 = B - A, bc0 = C - B, ca0 = A - C;
      const double pd = 0.5 * fmax(dot3(ab0, ab0), fmax(dot3(bc0, bc0), dot3(ca0, ca0))) + 1e-9;
      if (bx[0] > xh + pd || bx[3] < xl - pd || bx[1] > yh + pd || bx[4] < yl - pd || bx[2] > zh + pd || bx[5] < zl - pd) return;
      const int slot = atomicAdd(nbig, 1);
      if (slot < big_cap) {
        big[slot] = (int32_t)t;
        return;
      }
    }
    if (usable && E <= e_max) {
      const float pad = mpg_box_pad(E, pad_coef, pad_latlon, fmax(fabs(A.z), fmax(fabs(B.z), fabs(C.z))));
      const int i0 = max((int)ceilf(imin + di - pad), 0), i1 = min((int)floorf(imax + di + pad), npx - 1);
      const int j0 = max((int)ceilf(jmin + dj - pad), 0), j1 = min((int)floorf(jmax + dj + pad), npy - 1);
      if (i0 > i1 || j0 > j1) return;   // off the grid
      if ((i1 - i0 + 1) * (j1 - j0 + 1) <= 256) {
        dv3 ab = B - A, bc = C - B, ca = A - C;
        double e2 = fmax(dot3(ab, ab), fmax(dot3(bc, bc), dot3(ca, ca)));
        double pad3 = 0.5 * e2 + 1e-9;
        double lo[3] = {fmin(A.x, fmin(B.x, C.x)) - pad3, fmin(A.y, fmin(B.y, C.y)) - pad3, fmin(A.z, fmin(B.z, C.z)) - pad3};
        double hi[3] = {fmax(A.x, fmax(B.x, C.x)) + pad3, fmax(A.y, fmax(B.y, C.y)) + pad3, fmax(A.z, fmax(B.z, C.z)) + pad3};
        for (int j = j0; j <= j1; ++j)
          for (int i = i0; i <= i1; ++i) {
            int64_t p = (int64_t)j * npx + i;
            dv3 P = dv3{px[p], py[p], pz[p]};
            if (P.x < lo[0] || P.x > hi[0] || P.y < lo[1] || P.y > hi[1] || P.z < lo[2] || P.z > hi[2]) continue;
            double w[3];
            if (NORMAL ? tri_weights_normal(P, A, B, C, MPG_TOL, w) : tri_weights(P, A, B, C, MPG_TOL, w)) atomicMin(&owner[p], (int32_t)t);
          }
        return;
      }
    }
  }
  // AABB of the spherical triangle: planar AABB inflated by the bulge bound e^2/2 (e = longest edge)
  dv3 ab = B - A, bc = C - B, ca = A - C;
  double e2 = fmax(dot3(ab, ab), fmax(dot3(bc, bc), dot3(ca, ca)));
  double pad = 0.5 * e2 + 1e-9;
  double lo[3] = {fmin(A.x, fmin(B.x, C.x)) - pad, fmin(A.y, fmin(B.y, C.y)) - pad, fmin(A.z, fmin(B.z, C.z)) - pad};
  double hi[3] = {fmax(A.x, fmax(B.x, C.x)) + pad, fmax(A.y, fmax(B.y, C.y)) + pad, fmax(A.z, fmax(B.z, C.z)) + pad};

  // Depth-first walk.  A node's four children are box-tested BEFORE they are pushed -- their 4 x 6 bounds are fetched in one
  // round trip and only the ones the triangle's box meets go on the stack -- so the walk makes one iteration per node
  // that meets the triangle (one or two per level for a triangle smaller than a leaf), not four per level, each of which
  // was a dependent pop + box load (round 2; 82 % of this kernel's wave cycles were parked on those, profiles/r03_store_pmc.md).
  auto meets = [&](int lev, int node) -> bool {
    const double *bx = pyr.box + 6 * (pyr.off[lev] + node);
    return !(bx[0] > hi[0] || bx[3] < lo[0] || bx[1] > hi[1] || bx[4] < lo[1] || bx[2] > hi[2] || bx[5] < lo[2]);
  };
  // The node to visit next lives in a register (`cur`); only the SIBLINGS that also meet the triangle go on the stack, which
  // is a run-time-indexed private array, i.e. scratch memory: a triangle smaller than a leaf descends all ten levels without
  // touching it.
  int stack[RASTER_STACK];
  int sp = 0;
  int top = pyr.nlev - 1;
  int nleaf = 0;
  int cur = meets(top, 0) ? (top << 26) : -1;  // node 0 of the top level; node index < 2^26 per level
  for (;;) {
    if (cur < 0) {
      if (sp == 0) break;
      cur = stack[--sp];
    }
    const int e = cur;
    cur = -1;
    int lev = e >> 26;
    int node = e & ((1 << 26) - 1);
    int nxl = pyr.nx[lev];
    int bi = node % nxl, bj = node / nxl;
    if (lev == 0) {
      // A triangle that spreads over many leaves (the cells around the pole of a lat-lon grid cover thousands of points each;
      // every triangle of a mesh much coarser than the grid) is handed to k_tri_raster_big, one wavefront per triangle:
      // here it would keep one lane busy for milliseconds.  What this lane has already marked stays valid (atomicMin).
      if (++nleaf > RASTER_BIG_LEAVES && big_cap > 0) {
        int slot = atomicAdd(nbig, 1);
        if (slot < big_cap) {
          big[slot] = (int32_t)t;
          return;
        }
        big_cap = 0;   // queue full: finish here
      }
      int i0 = bi * MPG_PYR_B0, j0 = bj * MPG_PYR_B0;
      int i1 = min(i0 + MPG_PYR_B0, npx), j1 = min(j0 + MPG_PYR_B0, npy);
      for (int j = j0; j < j1; ++j)
        for (int i = i0; i < i1; ++i) {
          int64_t p = (int64_t)j * npx + i;
          dv3 P = dv3{px[p], py[p], pz[p]};
          if (P.x < lo[0] || P.x > hi[0] || P.y < lo[1] || P.y > hi[1] || P.z < lo[2] || P.z > hi[2]) continue;
          double w[3];
          if (NORMAL ? tri_weights_normal(P, A, B, C, MPG_TOL, w) : tri_weights(P, A, B, C, MPG_TOL, w)) atomicMin(&owner[p], (int32_t)t);
        }
    } else {
      int cnx = pyr.nx[lev - 1], cny = pyr.ny[lev - 1];
      bool go[4];
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        int ci = 2 * bi + (ch & 1), cj = 2 * bj + (ch >> 1);
        go[ch] = ci < cnx && cj < cny && meets(lev - 1, cj * cnx + ci);
      }
#pragma unroll
      for (int ch = 0; ch < 4; ++ch)
        if (go[ch]) {
          const int child = ((lev - 1) << 26) | ((2 * bj + (ch >> 1)) * cnx + 2 * bi + (ch & 1));
          // only nodes that meet the triangle are kept: a full stack cannot happen for a triangle smaller than the grid,
          // and if it ever did it is reported (MPG_ERR_OVERFLOW), never a silently unmapped point
          if (cur < 0) cur = child;
          else if (sp < RASTER_STACK) stack[sp++] = child;
          else atomicOr(overflow, 1);
        }
    }
  }
}

// One wavefront per handed-over triangle: the same walk with a per-wave stack in LDS; four lanes test a node's children,
// and a level-1 node (8 x 8 points; a lone level 0 of a tiny grid: 4 x 4) is tested one point per lane.
template <bool NORMAL>
__global__ __launch_bounds__(256) void k_tri_raster_big(const int32_t *__restrict__ big, const int32_t *__restrict__ nbig, int big_cap,
                                                        const int32_t *__restrict__ tri, int64_t triStride, const double *__restrict__ cx,
                                                        const double *__restrict__ cy, const double *__restrict__ cz, PyramidView pyr, int npx,
                                                        int npy, const double *__restrict__ px, const double *__restrict__ py,
                                                        const double *__restrict__ pz, int32_t *__restrict__ owner,
                                                        int32_t *__restrict__ overflow) {
  __shared__ int stk[4][RASTER_STACK];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n = min(*nbig, big_cap);
  for (int b = blockIdx.x * 4 + wv; b < n; b += gridDim.x * 4) {
    const int32_t t = big[b];
    const int32_t ia = tri[t], ib = tri[triStride + t], ic = tri[2 * triStride + t];
    const dv3 A = ld3(cx, cy, cz, ia), B = ld3(cx, cy, cz, ib), C = ld3(cx, cy, cz, ic);
    const dv3 ab = B - A, bc = C - B, ca = A - C;
    const double e2 = fmax(dot3(ab, ab), fmax(dot3(bc, bc), dot3(ca, ca)));
    const double pad = 0.5 * e2 + 1e-9;
    const double lo[3] = {fmin(A.x, fmin(B.x, C.x)) - pad, fmin(A.y, fmin(B.y, C.y)) - pad, fmin(A.z, fmin(B.z, C.z)) - pad};
    const double hi[3] = {fmax(A.x, fmax(B.x, C.x)) + pad, fmax(A.y, fmax(B.y, C.y)) + pad, fmax(A.z, fmax(B.z, C.z)) + pad};
    auto meets = [&](int lev, int node) -> bool {
      const double *bx = pyr.box + 6 * (pyr.off[lev] + node);
      return !(bx[0] > hi[0] || bx[3] < lo[0] || bx[1] > hi[1] || bx[4] < lo[1] || bx[2] > hi[2] || bx[5] < lo[2]);
    };
    int sp = 0;   // wave-uniform
    const int top = pyr.nlev - 1;
    if (meets(top, 0)) {
      if (lane == 0) stk[wv][0] = top << 26;
      sp = 1;
    }
    while (sp > 0) {
      const int e = stk[wv][--sp];
      const int lev = e >> 26, node = e & ((1 << 26) - 1);
      const int nxl = pyr.nx[lev];
      const int bi = node % nxl, bj = node / nxl;
      if (lev <= 1) {
        const int side = MPG_PYR_B0 << lev;
        const int i = bi * side + lane % side, j = bj * side + lane / side;
        if (lane < side * side && i < npx && j < npy) {
          const int64_t p = (int64_t)j * npx + i;
          const dv3 Pt = dv3{px[p], py[p], pz[p]};
          if (!(Pt.x < lo[0] || Pt.x > hi[0] || Pt.y < lo[1] || Pt.y > hi[1] || Pt.z < lo[2] || Pt.z > hi[2])) {
            double w[3];
            if (NORMAL ? tri_weights_normal(Pt, A, B, C, MPG_TOL, w) : tri_weights(Pt, A, B, C, MPG_TOL, w)) atomicMin(&owner[p], t);
          }
        }
      } else {
        const int cnx = pyr.nx[lev - 1], cny = pyr.ny[lev - 1];
        const int ci = 2 * bi + (lane & 1), cj = 2 * bj + ((lane >> 1) & 1);
        const bool go = lane < 4 && ci < cnx && cj < cny && meets(lev - 1, cj * cnx + ci);
        const unsigned long long m = __ballot(go);
        if (sp + 4 > RASTER_STACK) {
          if (lane == 0) atomicOr(overflow, 1);
        } else {
          if (go) stk[wv][sp + __popcll(m & ((1ull << lane) - 1))] = ((lev - 1) << 26) | (cj * cnx + ci);
          sp += __popcll(m);
        }
      }
    }
  }
}

// one thread per target point: weights of the owning triangle, SoA output
template <bool NORMAL>
__global__ __launch_bounds__(256) void k_tri_finalize(int64_t P, const int32_t *__restrict__ owner,
                                                      const int32_t *__restrict__ tri, int64_t triStride,
                                                      const double *__restrict__ cx, const double *__restrict__ cy,
                                                      const double *__restrict__ cz, const double *__restrict__ px,
                                                      const double *__restrict__ py, const double *__restrict__ pz,
                                                      int32_t *__restrict__ idx, double *__restrict__ w) {
  int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p >= P) return;
  int32_t t = owner[p];
  int32_t i0 = -1, i1 = -1, i2 = -1;
  double ww[3] = {0.0, 0.0, 0.0};
  if (t != 0x7fffffff) {
    i0 = tri[t];
    i1 = tri[triStride + t];
    i2 = tri[2 * triStride + t];
    dv3 Pt = dv3{px[p], py[p], pz[p]};
    if (NORMAL) tri_weights_normal(Pt, ld3(cx, cy, cz, i0), ld3(cx, cy, cz, i1), ld3(cx, cy, cz, i2), MPG_TOL, ww);
    else tri_weights(Pt, ld3(cx, cy, cz, i0), ld3(cx, cy, cz, i1), ld3(cx, cy, cz, i2), MPG_TOL, ww);
  }
  idx[p] = i0;
  idx[P + p] = i1;
  idx[2 * P + p] = i2;
  w[p] = ww[0];
  w[P + p] = ww[1];
  w[2 * P + p] = ww[2];
}

__global__ __launch_bounds__(256) void k_fill_i32(int64_t n, int32_t v, int32_t *__restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = v;
}

// Node-located sources (vorticity; interp.F90:350-366): the Voronoi polygons carry values at their
// corners.  ESMF triangulates polygons with > 4 sides in an undocumented order; this build uses the fan from the
// first listed vertex (triangle k of cell c = (v0, v_{k+1}, v_{k+2}), id = c*(maxEdges-2)+k) and the same
// rasteriser as the element-located case, with vertex coordinates as the source points.
__global__ __launch_bounds__(256) void k_fan_triangles(int64_t nCells, int maxEdges, int origin, const int32_t *__restrict__ voc,
                                                       const double *__restrict__ vx, const double *__restrict__ vy,
                                                       const double *__restrict__ vz, int32_t *__restrict__ ftri) {
  int nf = maxEdges - 2;
  int64_t nFan = nCells * nf;
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= nFan) return;
  int64_t c = t / nf;
  int k = (int)(t % nf);
  // the polygon's n listed (non-zero) vertices; apex = number (origin mod n) of them ("node_fan_origin": 0 the first, -1 the last),
  // the k-th fan triangle takes the (k+1)-th and (k+2)-th after the apex in listed order, cyclically
  int n = 0;
  for (int j = 0; j < maxEdges; ++j) n += voc[c * maxEdges + j] > 0;
  int32_t a = -1, b = -1, d = -1;
  if (k + 2 < n) {
    const int o = ((origin % n) + n) % n, ib = (o + k + 1) % n, id = (o + k + 2) % n;
    int m = 0;
    for (int j = 0; j < maxEdges; ++j) {
      int32_t x = voc[c * maxEdges + j];
      if (x <= 0) continue;
      if (m == o) a = x - 1;
      if (m == ib) b = x - 1;
      if (m == id) d = x - 1;
      ++m;
    }
  }
  if (d >= 0) {
    double det = det3_from(dv3{vx[a], vy[a], vz[a]}, dv3{vx[b], vy[b], vz[b]}, dv3{vx[d], vy[d], vz[d]});
    if (det == 0.0) a = b = d = -1;
    if (det < 0.0) { int32_t x = b; b = d; d = x; }
  } else {
    a = b = d = -1;
  }
  ftri[t] = a;
  ftri[nFan + t] = b;
  ftri[2 * nFan + t] = d;
}

int mpg_k_store_bilinear_mesh(mpg_mesh_s *m, mpg_grid_s *g, int stagger, int meshloc, mpg_handle_s *h, hipStream_t s) {
  int rc;
  // Triangles = those of the mesh's geometry window (everything unless the mesh was cut to this grid): the dual triangles of
  // its vertices, or the fan triangles of its cell rows; triangle NUMBERS are local to the window (their order -- lowest wins
  // on a shared edge -- is that of the global numbers), the ids INSIDE them are global.
  const int32_t *trip = m->tri.p;
  int64_t nT = m->vwn;
  const double *sx = m->cell.x.p, *sy = m->cell.y.p, *sz = m->cell.z.p;
  if (meshloc == MPG_MESHLOC_NODE) {
    nT = m->cwn * (int64_t)(m->maxEdges - 2);
    if (nT >= 0x7fffffff) {
      mpg_set_error("mesh too large for the node-located fan triangulation");
      return MPG_ERR_OVERFLOW;
    }
    sx = m->vx_g(); sy = m->vy_g(); sz = m->vz_g();
    const int origin = mpg_node_fan_origin();
    if ((!m->fan.p || m->fan_origin != origin) && nT > 0) {   // built once per mesh and apex rule
      if (!m->fan.p && (rc = m->fan.alloc(3 * (size_t)nT))) return rc;
      k_fan_triangles<<<(unsigned)((nT + 255) / 256), 256, 0, s>>>(m->cwn, m->maxEdges, origin, m->voc.p, sx, sy, sz, m->fan.p);
      m->fan_origin = origin;
    }
    trip = m->fan.p;
  }
  PointSet &pts = g->pts[stagger];
  int npx = g->snx[stagger], npy = g->sny[stagger];
  int64_t P = (int64_t)npx * npy;
  if (pts.n != P) {
    mpg_set_error("RegridStore: destination stagger %d has no coordinates", stagger);
    return MPG_ERR_INVALID_ARG;
  }
  if (!g->pyr[stagger].built && (rc = mpg_k_build_pyramid(pts, npx, npy, g->pyr[stagger], s))) return rc;
  if ((int64_t)g->pyr[stagger].nx[0] * g->pyr[stagger].ny[0] >= (1 << 26)) {
    mpg_set_error("target grid too large for the raster pyramid");
    return MPG_ERR_OVERFLOW;
  }
  h->kind = MPG_KIND_FIXED;
  h->nnz_per_row = 3;
  h->n_src = meshloc == MPG_MESHLOC_NODE ? m->nVertices : m->nCells;
  h->n_dst = P;
  h->nx_dst = npx;
  h->ny_dst = npy;
  h->nnz = 3 * P;
  if ((rc = h->idx.alloc(3 * (size_t)P))) return rc;
  if ((rc = h->w.alloc(3 * (size_t)P))) return rc;
  TmpBuf<int32_t> owner;
  if ((rc = owner.alloc((size_t)P, s))) return rc;
  int fb = (int)((P + 255) / 256);
  if (fb > 8192) fb = 8192;
  // one scratch allocation: [0] overflow flag, [1] length of the queue of handed-over triangles, [2 ...] the queue (nobody
  // reads the length on the host)
  const int big_cap = (int)std::min<int64_t>(nT, 1 << 18);   // 0 for an empty window: nothing is handed over
  TmpBuf<int32_t> ovf;
  if ((rc = ovf.alloc((size_t)big_cap + 2, s))) return rc;
  MPG_HIP(hipMemsetAsync(ovf.p, 0, 2 * sizeof(int32_t), s));
  k_fill_i32<<<fb, 256, 0, s>>>(P, 0x7fffffff, owner.p);
  auto raster = mpg_bilinear_linetype() ? k_tri_raster<true> : k_tri_raster<false>;
  auto raster_big = mpg_bilinear_linetype() ? k_tri_raster_big<true> : k_tri_raster_big<false>;
  auto finalize = mpg_bilinear_linetype() ? k_tri_finalize<true> : k_tri_finalize<false>;
  // a grid built from its projection: the source points' places in its index space (one inverse projection per point, shared by
  // the six triangles around it) give every triangle its candidate points in O(1); "store_boxes" 0 keeps the pyramid walk (A/B)
  TmpBuf<float> sij;
  const float *sijp = nullptr;
  float di = 0.f, dj = 0.f;
  if (mpg_grid_has_inverse(g, stagger) && mpg_store_boxes() && nT > 0) {
    const int64_t s0 = meshloc == MPG_MESHLOC_NODE ? m->vw0 : m->cw0, sn = meshloc == MPG_MESHLOC_NODE ? m->vwn : m->cwn;
    if ((rc = sij.alloc(2 * (size_t)sn, s))) return rc;
    if ((rc = mpg_k_points_ij(g, sn, sx + s0, sy + s0, sz + s0, sij.p, s))) return rc;
    sijp = sij.p - 2 * s0;   // indexed with global ids, like sx / sy / sz
    di = stagger == MPG_STAGGERLOC_EDGE1 ? 0.5f : 0.f;
    dj = stagger == MPG_STAGGERLOC_EDGE2 ? 0.5f : 0.f;
    h->store_path = 1;
  }
  if (nT > 0) {
    raster<<<(unsigned)((nT + 255) / 256), 256, 0, s>>>(nT, trip, nT, sx, sy, sz, mpg_pyr_view(g->pyr[stagger]), npx, npy, pts.x.p, pts.y.p, pts.z.p,
                                                       owner.p, ovf.p, ovf.p + 2, ovf.p + 1, big_cap, sijp, di, dj, (float)mpg_grid_box_pad_coef(g), (float)mpg_grid_box_pad_latlon(g),
                                                       (float)mpg_grid_box_emax(g));
    raster_big<<<1024, 256, 0, s>>>(ovf.p + 2, ovf.p + 1, big_cap, trip, nT, sx, sy, sz, mpg_pyr_view(g->pyr[stagger]), npx, npy, pts.x.p, pts.y.p,
                                   pts.z.p, owner.p, ovf.p);
  }
  finalize<<<(unsigned)((P + 255) / 256), 256, 0, s>>>(P, owner.p, trip, nT, sx, sy, sz, pts.x.p, pts.y.p, pts.z.p, h->idx.p, h->w.p);
  MPG_HIP(hipGetLastError());
  int32_t h2[2] = {0, 0};
  MPG_HIP(hipMemcpyAsync(h2, ovf.p, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  owner.free();
  const int32_t h_ovf = h2[0];
  h->store_stats[1] = std::min<int64_t>(h2[1], big_cap);   // triangles a wavefront rasterised (k_tri_raster_big)
  h->store_stats[2] = nT;
  if (h_ovf) {
    mpg_set_error("RegridStore(bilinear): traversal stack of the triangle rasteriser overflowed (pyramid deeper than %d levels)",
                  (RASTER_STACK - 1) / 3);
    return MPG_ERR_OVERFLOW;
  }
  return MPG_SUCCESS;
}

// mpg_init loads this translation unit's code object ahead of its first launch (mpg_api.hip: warm_modules)
const void *mpg_anchor_k_store_bilinear() { return (const void *)&k_tri_raster<false>; }
