// K6 "destagger_uv" weight generation: Grid -> Grid bilinear, CENTER -> EDGE1 / EDGE2.
//
// Replaces ESMF_FieldRegridStore(u_target_grid_nostag -> u_target_grid) and the V twin at
// interp.F90:298,316.  Semantics: source cells are the quads spanned by 4 neighbouring
// CENTER points; the weights come from the bilinear map X(xi,eta) = t*P solved by Newton in 3-D on the
// unit sphere; stagger points outside the hull of the centres (outer half-cell ring) are unmapped -> 0.
// The structured source needs no search: a U point (i-1/2, j) can only lie in quads (i-1, j-1) or
// (i-1, j); a V point (i, j-1/2) in quads (i-1, j-1) or (i, j-1).  Lowest quad id wins on shared edges.
// No floating-point contraction in this translation unit (see k_store_conserve.hip): what it computes -- weights, coordinates --
// is a function of the source text, not of which product the compiler chooses to fuse; explicit fma() calls stay what they are.
#pragma clang fp contract(off)
#include <math.h>

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
    // Newton converges quadratically: after a step below 1e-9 what is left is ~1e-18, far under the rounding noise of the
    // residual (1e-16 of a unit vector over a cell of 5e-4 rad = 2e-13 in xi / eta).  The former 1e-15 was below that
    // noise and never met: every point ran all 50 iterations (0.74 ms per stagger on configuration 4).
    if (fabs(ds) < 1e-9 && fabs(dt) < 1e-9) break;
  }
  *xi = s;
  *eta = t;
  return lam > 0.0;
}

// flags: MPG_GRID_* of the grid.  Periodic grids (ESMF_GridCreate1PeriDim + MONOPOLE, model_grid.F90:685-694):
// column a+1 wraps to 0, and rows -1 / ny-1 are the pole caps: triangles (pole, A, B) of the first / last CENTER
// row whose pole value is the row mean.  Quads are tried before caps.  pole_w / pole_dst / pole_src0 are
// [2][nxd]: slot 0 = south candidate row (j = 0), slot 1 = north candidate row (j = nyd-1).
__global__ __launch_bounds__(256) void k_grid_bilinear(int nx, int ny, int stagger, int flags, const double *__restrict__ cx,
                                                       const double *__restrict__ cy, const double *__restrict__ cz,
                                                       const double *__restrict__ px, const double *__restrict__ py,
                                                       const double *__restrict__ pz, int32_t *__restrict__ idx,
                                                       double *__restrict__ w, int32_t *__restrict__ pole_dst,
                                                       int32_t *__restrict__ pole_src0, double *__restrict__ pole_w, double tol) {
  int nxd = stagger == MPG_STAGGERLOC_EDGE1 ? nx + 1 : nx, nyd = stagger == MPG_STAGGERLOC_EDGE2 ? ny + 1 : ny;
  int64_t P = (int64_t)nxd * nyd;
  int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p >= P) return;
  const bool per = flags & MPG_GRID_PERIODIC_I;
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
      if (per) a = (a + nx) % nx;
      if (a < 0 || b < 0 || b + 1 >= ny || (!per && a + 1 >= nx)) continue;
      int a1 = a + 1 == nx ? 0 : a + 1;
      int64_t iA = (int64_t)b * nx + a, iB = (int64_t)b * nx + a1, iC = iB + nx, iD = iA + nx;
      double xi, eta;
      if (!quad_solve(Pt, ld3(cx, cy, cz, iA), ld3(cx, cy, cz, iB), ld3(cx, cy, cz, iC), ld3(cx, cy, cz, iD), &xi, &eta)) continue;
      if (xi < -tol || xi > 1.0 + tol || eta < -tol || eta > 1.0 + tol) continue;   // tol: MPG_TOL unless "grid_inside_tol_exp" says otherwise
      id[0] = (int32_t)iA; id[1] = (int32_t)iB; id[2] = (int32_t)iC; id[3] = (int32_t)iD;
      ww[0] = (1 - xi) * (1 - eta); ww[1] = xi * (1 - eta); ww[2] = xi * eta; ww[3] = (1 - xi) * eta;
      found = true;
    }
  double wp = 0.0;
  int32_t src0 = 0;
  if (per && !found)
    for (int bb = 0; bb < ncb && !found; ++bb)
      for (int aa = 0; aa < nca && !found; ++aa) {
        int b = cb[bb];
        bool south = b == -1 && !(flags & MPG_GRID_NO_SOUTH_POLE), north = b == ny - 1 && !(flags & MPG_GRID_NO_NORTH_POLE);
        if (!south && !north) continue;
        int a = (ca[aa] + nx) % nx, a1 = a + 1 == nx ? 0 : a + 1;
        int64_t row0 = south ? 0 : (int64_t)(ny - 1) * nx;
        dv3 A = ld3(cx, cy, cz, row0 + a), B = ld3(cx, cy, cz, row0 + a1);
        double t[3];
        // counter-clockwise seen from outside: (A, B, N) in the north, (B, A, S) in the south
        bool in = north ? tri_weights(Pt, A, B, dv3{0.0, 0.0, 1.0}, MPG_TOL, t) : tri_weights(Pt, B, A, dv3{0.0, 0.0, -1.0}, MPG_TOL, t);
        if (!in) continue;
        id[0] = (int32_t)(row0 + a); id[1] = (int32_t)(row0 + a1);
        id[2] = id[3] = id[0];  // zero-weight fillers: the 4-point Regrid kernel reads every slot of a mapped point
        ww[0] = north ? t[0] : t[1]; ww[1] = north ? t[1] : t[0];
        wp = t[2];
        src0 = (int32_t)row0;
        found = true;
      }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    idx[k * P + p] = id[k];
    w[k * P + p] = ww[k];
  }
  if (per && (j == 0 || j == nyd - 1)) {  // only these destination rows can touch a cap; unused slots stay zero
    int64_t q = (j == 0 ? 0 : nxd) + i;
    pole_dst[q] = (int32_t)p;
    pole_src0[q] = src0;
    pole_w[q] = wp;
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
  if (g->periodic & MPG_GRID_PERIODIC_I) {
    h->n_pole = 2 * (int64_t)npx;
    h->pole_len = g->nx;
    if ((rc = h->pole_dst.alloc(h->n_pole)) || (rc = h->pole_src0.alloc(h->n_pole)) || (rc = h->pole_w.alloc(h->n_pole))) return rc;
    MPG_HIP(hipMemsetAsync(h->pole_dst.p, 0, sizeof(int32_t) * h->n_pole, s));
    MPG_HIP(hipMemsetAsync(h->pole_src0.p, 0, sizeof(int32_t) * h->n_pole, s));
    MPG_HIP(hipMemsetAsync(h->pole_w.p, 0, sizeof(double) * h->n_pole, s));
  }
  k_grid_bilinear<<<(unsigned)((P + 255) / 256), 256, 0, s>>>(g->nx, g->ny, dst_stagger, g->periodic, cen.x.p, cen.y.p, cen.z.p,
                                                             dst.x.p, dst.y.p, dst.z.p, h->idx.p, h->w.p, h->pole_dst.p,
                                                             h->pole_src0.p, h->pole_w.p, mpg_grid_inside_tol_exp() == 10 ? MPG_TOL : pow(10.0, -(double)mpg_grid_inside_tol_exp()));
  MPG_HIP(hipGetLastError());
  MPG_HIP(hipStreamSynchronize(s));
  return MPG_SUCCESS;
}

// mpg_init loads this translation unit's code object ahead of its first launch (mpg_api.hip: warm_modules)
const void *mpg_anchor_k_store_gridbil() { return (const void *)k_grid_bilinear; }
