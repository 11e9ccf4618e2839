// K4 "build_conserve_csr": first-order conservative RegridStore.
//
// Replaces ESMF_Field[Bundle]RegridStore(regridmethod=CONSERVE) at interp.F90:372,394 (snow, snowh;
// input_data.F90:840).  Semantics: w_ij = Area(src_i ^ dst_j) / Area(dst_j)
// (normType=DSTAREA) with great-circle polygon sides on the unit sphere; src polygon = Voronoi cell
// from verticesOnCell/vertex coordinates, dst polygon = the 4 CORNER-stagger points around centre
// (i,j) (model_grid.F90:784-794,959-984).  Uncovered destination cells stay 0.
//
// MI355X-native formulation: source polygons are rasterised onto the destination cells through an
// AABB pyramid over the CORNER points (same machinery as the bilinear rasteriser).  Pass 1 counts the
// overlaps per destination cell, a rocPRIM scan turns counts into CSR row offsets, pass 2 recomputes
// and fills, pass 3 sorts every (short) row by source id so the stored matrix and the summation order
// are deterministic.  Clipping = Sutherland-Hodgman against the 4 great-circle half-spaces.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "geom.h"
#include "mpg_internal.h"

#define CONS_MAXV 12   // max source polygon vertices handled (MPAS maxEdges is 6..10)
#define CONS_BUF (CONS_MAXV + 6)
#define CONS_STACK 64

__device__ int clip_halfspace(int n, const dv3 *in, dv3 nrm, dv3 *out) {
  int m = 0;
  double eps = 1e-15 * sqrt(dot3(nrm, nrm));
  for (int i = 0; i < n; ++i) {
    dv3 X1 = in[i], X2 = in[(i + 1 == n) ? 0 : i + 1];
    double d1 = dot3(nrm, X1), d2 = dot3(nrm, X2);
    bool in1 = d1 >= -eps, in2 = d2 >= -eps;
    if (in1 && m < CONS_BUF) out[m++] = X1;
    if (in1 != in2 && m < CONS_BUF) {
      dv3 X = X1 * d2 - X2 * d1;
      double sgn = (d2 - d1) > 0.0 ? 1.0 : -1.0;
      double nn = sqrt(dot3(X, X));
      if (nn > 0.0) out[m++] = X * (sgn / nn);
    }
  }
  return m;
}
__device__ double clip_area(int ns, const dv3 *src, const dv3 *quad) {
  dv3 a[CONS_BUF], b[CONS_BUF];
  int n = ns;
  for (int i = 0; i < ns; ++i) a[i] = src[i];
  dv3 *cur = a, *nxt = b;
  for (int e = 0; e < 4 && n >= 3; ++e) {
    // a collapsed side (the two CORNER points of a lat-lon cell at a pole, equal up to the rounding of cos(90))
    // bounds nothing, and the direction of its great circle is noise: skip it
    dv3 side = quad[(e + 1) & 3] - quad[e];
    if (dot3(side, side) < 1e-24) continue;
    dv3 nrm = cross3(quad[e], quad[(e + 1) & 3]);
    n = clip_halfspace(n, cur, nrm, nxt);
    dv3 *t = cur; cur = nxt; nxt = t;
  }
  if (n < 3) return 0.0;
  double s = 0.0;
  for (int i = 1; i + 1 < n; ++i) s += sph_tri_area(cur[0], cur[i], cur[i + 1]);
  return s > 0.0 ? s : 0.0;
}

// FILL = false: count[p]++ ; FILL = true: write (col, val) at rowptr[p] + cursor[p]++
template <bool FILL>
__global__ __launch_bounds__(128) void k_conserve_raster(int64_t nCells, int maxEdges, const int32_t *__restrict__ voc,
                                                         const double *__restrict__ vx, const double *__restrict__ vy,
                                                         const double *__restrict__ vz, PyramidView pyr, int nx, int ny,
                                                         const double *__restrict__ qx, const double *__restrict__ qy,
                                                         const double *__restrict__ qz, int32_t *__restrict__ count,
                                                         const int32_t *__restrict__ rowptr, int32_t *__restrict__ col,
                                                         double *__restrict__ val) {
  int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= nCells) return;
  dv3 poly[CONS_MAXV];
  int n = 0;
  for (int j = 0; j < maxEdges && n < CONS_MAXV; ++j) {
    int32_t v = voc[c * maxEdges + j];
    if (v > 0) poly[n++] = dv3{vx[v - 1], vy[v - 1], vz[v - 1]};
  }
  if (n < 3) return;
  double area = 0.0;
  for (int i = 1; i + 1 < n; ++i) area += sph_tri_area(poly[0], poly[i], poly[i + 1]);
  if (area == 0.0) return;
  if (area < 0.0)  // make CCW seen from outside
    for (int i = 0; i < n / 2; ++i) {
      dv3 t = poly[i]; poly[i] = poly[n - 1 - i]; poly[n - 1 - i] = t;
    }
  double lo[3] = {2, 2, 2}, hi[3] = {-2, -2, -2}, e2 = 0.0;
  for (int i = 0; i < n; ++i) {
    lo[0] = fmin(lo[0], poly[i].x); hi[0] = fmax(hi[0], poly[i].x);
    lo[1] = fmin(lo[1], poly[i].y); hi[1] = fmax(hi[1], poly[i].y);
    lo[2] = fmin(lo[2], poly[i].z); hi[2] = fmax(hi[2], poly[i].z);
    dv3 d = poly[i] - poly[0];
    e2 = fmax(e2, dot3(d, d));
  }
  double pad = 2.0 * e2 + 1e-9;  // bulge of a polygon of diameter <= 2*sqrt(e2)
#pragma unroll
  for (int k = 0; k < 3; ++k) { lo[k] -= pad; hi[k] += pad; }

  int nxc = nx + 1;
  int stack[CONS_STACK];
  int sp = 0;
  stack[sp++] = (pyr.nlev - 1) << 26;
  while (sp > 0) {
    int e = stack[--sp];
    int lev = e >> 26, node = e & ((1 << 26) - 1);
    const double *bx = pyr.box + 6 * (pyr.off[lev] + node);
    // node boxes of the cell pyramid already include the destination cells' own bulge (k_pyr_leaf, halo mode)
    if (bx[0] > hi[0] || bx[3] < lo[0] || bx[1] > hi[1] || bx[4] < lo[1] || bx[2] > hi[2] || bx[5] < lo[2]) continue;
    int nxl = pyr.nx[lev];
    int bi = node % nxl, bj = node / nxl;
    if (lev == 0) {
      int i0 = bi * MPG_PYR_B0, j0 = bj * MPG_PYR_B0;
      int i1 = min(i0 + MPG_PYR_B0, nx), j1 = min(j0 + MPG_PYR_B0, ny);
      for (int j = j0; j < j1; ++j)
        for (int i = i0; i < i1; ++i) {
          int64_t k00 = (int64_t)j * nxc + i;
          dv3 q[4] = {dv3{qx[k00], qy[k00], qz[k00]}, dv3{qx[k00 + 1], qy[k00 + 1], qz[k00 + 1]},
                      dv3{qx[k00 + nxc + 1], qy[k00 + nxc + 1], qz[k00 + nxc + 1]}, dv3{qx[k00 + nxc], qy[k00 + nxc], qz[k00 + nxc]}};
          double ql[3] = {2, 2, 2}, qh[3] = {-2, -2, -2}, qe2 = 0.0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            ql[0] = fmin(ql[0], q[k].x); qh[0] = fmax(qh[0], q[k].x);
            ql[1] = fmin(ql[1], q[k].y); qh[1] = fmax(qh[1], q[k].y);
            ql[2] = fmin(ql[2], q[k].z); qh[2] = fmax(qh[2], q[k].z);
            dv3 d = q[k] - q[0];
            qe2 = fmax(qe2, dot3(d, d));
          }
          double qp = 2.0 * qe2 + 1e-9;
          if (ql[0] - qp > hi[0] || qh[0] + qp < lo[0] || ql[1] - qp > hi[1] || qh[1] + qp < lo[1] || ql[2] - qp > hi[2] || qh[2] + qp < lo[2]) continue;
          double aq = sph_tri_area(q[0], q[1], q[2]) + sph_tri_area(q[0], q[2], q[3]);
          if (aq < 0.0) { dv3 t = q[1]; q[1] = q[3]; q[3] = t; aq = -aq; }
          if (!(aq > 0.0)) continue;
          double ar = clip_area(n, poly, q);
          if (ar > 1e-14 * aq) {
            int64_t p = (int64_t)j * nx + i;
            int slot = atomicAdd(&count[p], 1);
            if (FILL) {
              col[rowptr[p] + slot] = (int32_t)c;
              val[rowptr[p] + slot] = ar / aq;
            }
          }
        }
    } else {
      int cnx = pyr.nx[lev - 1], cny = pyr.ny[lev - 1];
#pragma unroll
      for (int dj = 0; dj < 2; ++dj)
#pragma unroll
        for (int di = 0; di < 2; ++di) {
          int ci = 2 * bi + di, cj = 2 * bj + dj;
          if (ci < cnx && cj < cny && sp < CONS_STACK) stack[sp++] = ((lev - 1) << 26) | (cj * cnx + ci);
        }
    }
  }
}

__global__ __launch_bounds__(256) void k_csr_sort_rows(int64_t P, const int32_t *__restrict__ rowptr, int32_t *__restrict__ col,
                                                       double *__restrict__ val) {
  int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p >= P) return;
  int b = rowptr[p], e = rowptr[p + 1];
  for (int i = b + 1; i < e; ++i) {
    int32_t kc = col[i];
    double kv = val[i];
    int j = i - 1;
    while (j >= b && col[j] > kc) {
      col[j + 1] = col[j];
      val[j + 1] = val[j];
      --j;
    }
    col[j + 1] = kc;
    val[j + 1] = kv;
  }
}

int mpg_k_store_conserve(mpg_mesh_s *m, mpg_grid_s *g, mpg_handle_s *h, hipStream_t s) {
  int rc;
  PointSet &cor = g->pts[MPG_STAGGERLOC_CORNER];
  int nx = g->nx, ny = g->ny;
  int64_t P = (int64_t)nx * ny;
  if (cor.n != (int64_t)(nx + 1) * (ny + 1)) {
    mpg_set_error("conservative RegridStore needs CORNER-stagger coordinates on the destination grid");
    return MPG_ERR_INVALID_ARG;
  }
  if (m->maxEdges > CONS_MAXV) {
    mpg_set_error("conservative RegridStore: maxEdges %d > %d", m->maxEdges, CONS_MAXV);
    return MPG_ERR_UNSUPPORTED;
  }
  if (!g->cellpyr.built && (rc = mpg_k_build_cell_pyramid(cor, nx, ny, g->cellpyr, s))) return rc;
  h->kind = MPG_KIND_CSR;
  h->nnz_per_row = 0;
  h->n_src = m->nCells;
  h->n_dst = P;
  h->nx_dst = nx;
  h->ny_dst = ny;
  TmpBuf<int32_t> count;
  if ((rc = count.alloc((size_t)P + 1)) || (rc = h->rowptr.alloc((size_t)P + 1))) return rc;
  MPG_HIP(hipMemsetAsync(count.p, 0, sizeof(int32_t) * (P + 1), s));
  unsigned nb = (unsigned)((m->nCells + 127) / 128);
  PyramidView pv = mpg_pyr_view(g->cellpyr);
  k_conserve_raster<false><<<nb, 128, 0, s>>>(m->nCells, m->maxEdges, m->voc.p, m->vert.x.p, m->vert.y.p, m->vert.z.p, pv, nx, ny,
                                            cor.x.p, cor.y.p, cor.z.p, count.p, nullptr, nullptr, nullptr);
  MPG_HIP(hipGetLastError());
  size_t tmp_bytes = 0;
  MPG_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, count.p, h->rowptr.p, (int32_t)0, (size_t)P + 1, rocprim::plus<int32_t>(), s));
  TmpBuf<char> tmp;
  if ((rc = tmp.alloc(tmp_bytes + 16))) return rc;
  MPG_HIP(rocprim::exclusive_scan((void *)tmp.p, tmp_bytes, count.p, h->rowptr.p, (int32_t)0, (size_t)P + 1, rocprim::plus<int32_t>(), s));
  int32_t nnz = 0;
  MPG_HIP(hipMemcpyAsync(&nnz, h->rowptr.p + P, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  if (nnz < 0) {
    mpg_set_error("conservative weight matrix exceeds 2^31 entries");
    return MPG_ERR_OVERFLOW;
  }
  h->nnz = nnz;
  if ((rc = h->col.alloc((size_t)nnz + 1)) || (rc = h->val.alloc((size_t)nnz + 1))) return rc;
  MPG_HIP(hipMemsetAsync(count.p, 0, sizeof(int32_t) * (P + 1), s));
  k_conserve_raster<true><<<nb, 128, 0, s>>>(m->nCells, m->maxEdges, m->voc.p, m->vert.x.p, m->vert.y.p, m->vert.z.p, pv, nx, ny,
                                           cor.x.p, cor.y.p, cor.z.p, count.p, h->rowptr.p, h->col.p, h->val.p);
  k_csr_sort_rows<<<(unsigned)((P + 255) / 256), 256, 0, s>>>(P, h->rowptr.p, h->col.p, h->val.p);
  MPG_HIP(hipGetLastError());
  MPG_HIP(hipStreamSynchronize(s));
  count.free();
  tmp.free();
  return MPG_SUCCESS;
}
