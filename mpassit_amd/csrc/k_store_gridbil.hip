// K6 "destagger_uv" weight generation: Grid -> Grid bilinear, CENTER -> EDGE1 / EDGE2.
//
// Replaces ESMF_FieldRegridStore(u_target_grid_nostag -> u_target_grid) and the V twin at
// interp.F90:298,316.  Semantics: source cells are the quads spanned by 4 neighbouring
// CENTER points; the weights come from the bilinear map X(xi,eta) = t*P solved by Newton in 3-D on the
// unit sphere; stagger points outside the hull of the centres (outer half-cell ring) are unmapped -> 0.
// The structured source needs no search: a U point (i-1/2, j) can only lie in quads (i-1, j-1) or
// (i-1, j); a V point (i, j-1/2) in quads (i-1, j-1) or (i, j-1).  Lowest quad id wins on shared edges.
#include "geom.h"
#include "mpg_internal.h"

__device__ bool quad_solve(dv3 P, dv3 A, dv3 B, dv3 C, dv3 D, double *xi, double *eta) {
  double s = 0.5, t = 0.5, lam = 1.0;
  dv3 e1 = B - A, e2 = D - A, e3 = (A - B) + (C - D);
  for (int it = 0; it < 50; ++it) {
    dv3 X = (A + e1 * s) + (e2 * t + e3 * (s * t));
    dv3 F = X - P * lam;
    dv3 Js = e1 + e3 * t, Jt = e2 + e3 * s, Jl = P * -1.0;
    double det = dot3(Js, cross3(Jt, Jl));
    if (det == 0.0) return false;
    dv3 mF = F * -1.0;
    double ds = dot3(mF, cross3(Jt, Jl)) / det;
    double dt = dot3(Js, cross3(mF, Jl)) / det;
    double dl = dot3(Js, cross3(Jt, mF)) / det;
    s += ds;
    t += dt;
    lam += dl;
    if (fabs(ds) < 1e-15 && fabs(dt) < 1e-15) break;
  }
  *xi = s;
  *eta = t;
  return lam > 0.0;
}

__global__ __launch_bounds__(256) void k_grid_bilinear(int nx, int ny, int stagger, const double *__restrict__ cx,
                                                       const double *__restrict__ cy, const double *__restrict__ cz,
                                                       const double *__restrict__ px, const double *__restrict__ py,
                                                       const double *__restrict__ pz, int32_t *__restrict__ idx,
                                                       double *__restrict__ w) {
  int nxd = stagger == MPG_STAGGERLOC_EDGE1 ? nx + 1 : nx, nyd = stagger == MPG_STAGGERLOC_EDGE2 ? ny + 1 : ny;
  int64_t P = (int64_t)nxd * nyd;
  int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p >= P) return;
  int i = (int)(p % nxd), j = (int)(p / nxd);
  dv3 Pt = dv3{px[p], py[p], pz[p]};
  int ca[2], cb[2], nca, ncb;
  if (stagger == MPG_STAGGERLOC_EDGE1) { ca[0] = i - 1; nca = 1; cb[0] = j - 1; cb[1] = j; ncb = 2; }
  else { ca[0] = i - 1; ca[1] = i; nca = 2; cb[0] = j - 1; ncb = 1; }
  int32_t id[4] = {-1, -1, -1, -1};
  double ww[4] = {0, 0, 0, 0};
  bool found = false;
  for (int bb = 0; bb < ncb && !found; ++bb)
    for (int aa = 0; aa < nca && !found; ++aa) {
      int a = ca[aa], b = cb[bb];
      if (a < 0 || b < 0 || a + 1 >= nx || b + 1 >= ny) continue;
      int64_t iA = (int64_t)b * nx + a, iB = iA + 1, iC = iA + nx + 1, iD = iA + nx;
      double xi, eta;
      if (!quad_solve(Pt, ld3(cx, cy, cz, iA), ld3(cx, cy, cz, iB), ld3(cx, cy, cz, iC), ld3(cx, cy, cz, iD), &xi, &eta)) continue;
      if (xi < -MPG_TOL || xi > 1.0 + MPG_TOL || eta < -MPG_TOL || eta > 1.0 + MPG_TOL) continue;
      id[0] = (int32_t)iA; id[1] = (int32_t)iB; id[2] = (int32_t)iC; id[3] = (int32_t)iD;
      ww[0] = (1 - xi) * (1 - eta); ww[1] = xi * (1 - eta); ww[2] = xi * eta; ww[3] = (1 - xi) * eta;
      found = true;
    }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    idx[k * P + p] = id[k];
    w[k * P + p] = ww[k];
  }
}

int mpg_k_store_grid_bilinear(mpg_grid_s *g, int dst_stagger, mpg_handle_s *h, hipStream_t s) {
  int rc;
  PointSet &cen = g->pts[MPG_STAGGERLOC_CENTER];
  PointSet &dst = g->pts[dst_stagger];
  int npx = g->snx[dst_stagger], npy = g->sny[dst_stagger];
  int64_t P = (int64_t)npx * npy;
  if (dst.n != P || cen.n != (int64_t)g->nx * g->ny) {
    mpg_set_error("grid RegridStore: stagger %d has no coordinates", dst_stagger);
    return MPG_ERR_INVALID_ARG;
  }
  h->kind = MPG_KIND_FIXED;
  h->nnz_per_row = 4;
  h->n_src = cen.n;
  h->n_dst = P;
  h->nx_dst = npx;
  h->ny_dst = npy;
  h->nnz = 4 * P;
  if ((rc = h->idx.alloc(4 * (size_t)P)) || (rc = h->w.alloc(4 * (size_t)P))) return rc;
  k_grid_bilinear<<<(unsigned)((P + 255) / 256), 256, 0, s>>>(g->nx, g->ny, dst_stagger, cen.x.p, cen.y.p, cen.z.p, dst.x.p,
                                                             dst.y.p, dst.z.p, h->idx.p, h->w.p);
  MPG_HIP(hipGetLastError());
  MPG_HIP(hipStreamSynchronize(s));
  return MPG_SUCCESS;
}
