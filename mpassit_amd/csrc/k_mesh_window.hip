// mpg_mesh_create_window: the part of an MPAS mesh one rank's target grid can see.
//
// The reference gives every rank 1/N of the cells (para_range, model_grid.F90:423-438, 2428-2441) and lets ESMF find out
// which rank's elements a destination point needs.  Here a rank's Stores only ever look at the cells near ITS rows of the
// target grid, so a rank brings only those to the device: connectivity (verticesOnCell), vertex coordinates, dual
// triangles and the nearest-neighbour BVH are built for a WINDOW of the mesh, ids stay global (handles, source ranges
// and the halo schedule do not change).  Round 3 measured why: with the rows split over 8 ranks a rank's three Stores take
// 2 ms, the whole-mesh geometry ingest beside them 4.5-12.8 ms at any N.
//
// How the window is cut, and why the Stores on it give the very weights of the whole mesh:
//   1. every cell CENTRE goes to the device (16 B per cell: the one part that does not shrink with N) and gets its distance
//      D to the grid -- to the union of the leaf boxes of the grid's pyramids (k_setup.hip): the 4 x 4-point boxes of the
//      point staggers it has, the padded 4 x 4-cell boxes of its CORNER mesh.  Cells with D <= delta are SELECTED;
//   2. the window's cell rows are the covering id range [c0, c1) of the selected cells (with spatially banded numbering --
//      what MPAS meshes and the driver's source windows already rely on -- barely more than the selection; with arbitrary
//      numbering up to the whole mesh: slower, never wrong), its vertices the covering range of what those rows reference;
//   3. CLOSURE is then verified on the device, not assumed.  A vertex of the window with fewer than three cells among the
//      window's rows is either on the mesh's rim or misses a cell that lies outside the window.  With r = |cell - vertex|
//      and the one assumption that the cells around a vertex are within a factor K = 2 of each other's distance from it
//      (1 for a Voronoi mesh: the vertex is the circumcentre), a missing cell b of vertex v next to a known cell a has
//      |b - a| <= (1 + K) r, so it would have been SELECTED if D(a) + (1 + K) r <= delta: such a vertex is PROVEN to be rim.
//      Every cell whose dual triangles or whose own polygon can reach the grid -- D(a) <= (1 + K) r_a, r_a its largest
//      vertex distance -- must have all its vertices complete or proven rim (a cell beyond the distance pass's cap of 2 delta that
//      is LARGE enough to reach the grid from there -- a coarse cell of a variable-resolution mesh -- has its distance measured for
//      this test).  If one is not, delta doubles and the window is cut again (six times at most, then the whole mesh is taken);
//   4. nearest-neighbour search needs no closure: a cell outside the selection is further than delta from every grid point,
//      so any point whose nearest SITE is within delta has its true answer, and mpg_k_store_nearest checks exactly that
//      (k_store_nearest.hip), falling back to a BVH over all centres, which are on the device anyway.
// Triangle and cell numbers inside the window keep the order of the global ones, so "lowest id wins" (shared edges of the
// rasteriser, ties of the search, the sorted rows of the conservative matrix) resolves as on the whole mesh: the weights are
// bit-identical (tests/test_mesh_window_gpu.py asks for array_equal on every method).
#include <math.h>
#include <string.h>

#include "geom.h"
#include "mpg_internal.h"

#define WIN_K 2.0        // assumed bound on the ratio of two cells' distances from a vertex they share
#define WIN_STACK 64
#define WIN_MAXPYR 4

struct WinPyrs {
  int n;
  PyramidView v[WIN_MAXPYR];
};

// chord length between neighbouring CENTER points at 3 x 3 sample positions, the largest of them
__global__ void k_grid_spacing(int nx, int ny, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                               unsigned long long *__restrict__ out) {
  const int t = threadIdx.x;   // 18 samples: 9 positions x 2 directions
  if (t >= 18) return;
  const int pos = t >> 1, dir = t & 1;
  int i = (nx - 1) * (pos % 3) / 2, j = (ny - 1) * (pos / 3) / 2;
  int i2 = i + (dir == 0), j2 = j + (dir == 1);
  if (i2 >= nx) { i2 = i; i = i > 0 ? i - 1 : 0; }
  if (j2 >= ny) { j2 = j; j = j > 0 ? j - 1 : 0; }
  const int64_t p = (int64_t)j * nx + i, q = (int64_t)j2 * nx + i2;
  const double d = sqrt(dist2_nofma(x[p], y[p], z[p], x[q], y[q], z[q]));
  atomicMax(out, (unsigned long long)__double_as_longlong(d));
}

// squared distance from (X, Y, Z) to the nearest leaf box of one pyramid if that is <= best2 (found = true), else best2
__device__ double pyr_dist2(const PyramidView &pyr, double X, double Y, double Z, double best2, bool &found) {
  int stack[WIN_STACK];
  int sp = 0;
  const int top = pyr.nlev - 1;
  if (boxdist2_nofma(X, Y, Z, pyr.box + 6 * pyr.off[top]) > best2) return best2;
  stack[sp++] = top << 26;
  while (sp > 0) {
    const int e = stack[--sp];
    const int lev = e >> 26, node = e & ((1 << 26) - 1);
    const double d = boxdist2_nofma(X, Y, Z, pyr.box + 6 * (pyr.off[lev] + node));
    if (d > best2) continue;
    if (lev == 0) {
      best2 = d;
      found = true;
      if (d == 0.0) break;
      continue;
    }
    const int nxl = pyr.nx[lev], bi = node % nxl, bj = node / nxl;
    const int cnx = pyr.nx[lev - 1], cny = pyr.ny[lev - 1];
    // children within the bound, the nearest pushed last (popped first)
    double cd[4];
    int cc[4], nc = 0;
    for (int ch = 0; ch < 4; ++ch) {
      const int ci = 2 * bi + (ch & 1), cj = 2 * bj + (ch >> 1);
      if (ci >= cnx || cj >= cny) continue;
      const int child = cj * cnx + ci;
      const double dc = boxdist2_nofma(X, Y, Z, pyr.box + 6 * (pyr.off[lev - 1] + child));
      if (dc > best2) continue;
      int k = nc++;
      while (k > 0 && cd[k - 1] < dc) {
        cd[k] = cd[k - 1];
        cc[k] = cc[k - 1];
        --k;
      }
      cd[k] = dc;
      cc[k] = child;
    }
    for (int k = 0; k < nc && sp < WIN_STACK; ++k) stack[sp++] = ((lev - 1) << 26) | cc[k];
  }
  return best2;
}

// D[c] = distance of cell c to the grid (to the nearest leaf box of any of its pyramids), INFINITY beyond `cap`.
// stats: [0] cells with D == 0, [1] cells with D <= delta, [2] their lowest id, [3] their highest id + 1
__global__ __launch_bounds__(256) void k_cell_dist(int64_t n, const double *__restrict__ cx, const double *__restrict__ cy,
                                                   const double *__restrict__ cz, WinPyrs pyrs, double cap, double delta, float *__restrict__ D,
                                                   unsigned long long *__restrict__ stats) {
  __shared__ unsigned long long s_in, s_sel, s_lo, s_hi;
  if (threadIdx.x == 0) {
    s_in = 0; s_sel = 0; s_lo = ~0ull; s_hi = 0;
  }
  __syncthreads();
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c < n) {
    const double X = cx[c], Y = cy[c], Z = cz[c];
    const double cap2 = cap * cap;
    double best2 = cap2;
    bool found = false;
    for (int q = 0; q < pyrs.n && !(found && best2 == 0.0); ++q) best2 = pyr_dist2(pyrs.v[q], X, Y, Z, best2, found);
    const double d = found ? sqrt(best2) : INFINITY;
    if (D) D[c] = found ? (float)d : INFINITY;
    if (found && d == 0.0) atomicAdd(&s_in, 1ull);
    if (found && d <= delta) {
      atomicAdd(&s_sel, 1ull);
      atomicMin(&s_lo, (unsigned long long)c);
      atomicMax(&s_hi, (unsigned long long)c + 1);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_in) atomicAdd(stats + 0, s_in);
    if (s_sel) {
      atomicAdd(stats + 1, s_sel);
      atomicMin(stats + 2, s_lo);
      atomicMax(stats + 3, s_hi);
    }
  }
}

// lowest / highest + 1 vertex id (0-based) the rows of `voc` reference
__global__ __launch_bounds__(256) void k_vertex_range(int64_t nent, const int32_t *__restrict__ voc, unsigned long long *__restrict__ out) {
  __shared__ unsigned long long s_lo, s_hi;
  if (threadIdx.x == 0) {
    s_lo = ~0ull; s_hi = 0;
  }
  __syncthreads();
  unsigned long long lo = ~0ull, hi = 0;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < nent; e += (int64_t)gridDim.x * blockDim.x) {
    const int32_t v = voc[e];
    if (v > 0) {
      lo = lo < (unsigned long long)(v - 1) ? lo : (unsigned long long)(v - 1);
      hi = hi > (unsigned long long)v ? hi : (unsigned long long)v;
    }
  }
  if (hi) {
    atomicMin(&s_lo, lo);
    atomicMax(&s_hi, hi);
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_hi) {
    atomicMin(out, s_lo);
    atomicMax(out + 1, s_hi);
  }
}

// closure, step 1: status of every vertex of the window -- 0 complete (three cells among the window's rows, or unused),
// 1 proven to be on the mesh's rim, 2 open (a cell outside the window may touch it).  tri holds the raw slots of k_tri_scatter.
__global__ __launch_bounds__(256) void k_win_vertex_status(int64_t nV, const int32_t *__restrict__ cnt, const int32_t *__restrict__ tri,
                                                           const double *__restrict__ cx, const double *__restrict__ cy, const double *__restrict__ cz,
                                                           const double *__restrict__ vx, const double *__restrict__ vy, const double *__restrict__ vz,
                                                           const float *__restrict__ D, double delta, uint8_t *__restrict__ status) {
  const int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (v >= nV) return;
  const int n = cnt[v];
  uint8_t st = 0;
  if (n == 1 || n == 2) {
    st = 2;
    for (int k = 0; k < n; ++k) {
      const int32_t a = tri[(int64_t)k * nV + v];
      const double r = sqrt(dist2_nofma(cx[a], cy[a], cz[a], vx[v], vy[v], vz[v]));
      if ((double)D[a] * (1.0 + 1e-6) + (1.0 + WIN_K) * r <= delta) st = 1;   // a third cell would lie within delta of the grid: it would be here
    }
  }
  status[v] = st;
}
// closure, step 2: a cell whose triangles or polygon can reach the grid must have no open vertex.  voc: the window's rows.
__global__ __launch_bounds__(256) void k_win_cell_closed(int64_t nC, int64_t cell0, int64_t vert0, int maxEdges, const int32_t *__restrict__ voc,
                                                         const double *__restrict__ cx, const double *__restrict__ cy, const double *__restrict__ cz,
                                                         const double *__restrict__ vx, const double *__restrict__ vy, const double *__restrict__ vz,
                                                         const float *__restrict__ D, const uint8_t *__restrict__ status, int32_t *__restrict__ bad,
                                                         WinPyrs pyrs, double dcap) {
  const int64_t cl = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (cl >= nC) return;
  const int64_t a = cell0 + cl;
  double d = (double)D[a];   // INFINITY beyond dcap = 2 delta: measured again below for a cell large enough to reach the grid from there
  double rmax = 0.0;
  bool open = false;
  for (int j = 0; j < maxEdges; ++j) {
    const int32_t v1 = voc[cl * maxEdges + j];
    if (v1 <= 0) continue;
    const int64_t v = v1 - 1 - vert0;
    rmax = fmax(rmax, sqrt(dist2_nofma(cx[a], cy[a], cz[a], vx[v], vy[v], vz[v])));
    open = open || status[v] == 2;
  }
  if (!open) return;
  if (!(d < INFINITY)) {
    // Beyond the cap of the distance pass.  On a quasi-uniform mesh such a cell is a dozen spacings from the grid and cannot reach
    // it; on a VARIABLE-resolution mesh the window's id range may hold a coarse cell whose own size exceeds the margin that was cut
    // for the fine cells near the grid (round 5, advisor): its distance is taken for real, up to what it could span
    const double reach = (1.0 + WIN_K) * rmax;
    if (reach <= dcap) return;
    double best2 = reach * reach;
    bool found = false;
    for (int q = 0; q < pyrs.n && !(found && best2 == 0.0); ++q) best2 = pyr_dist2(pyrs.v[q], cx[a], cy[a], cz[a], best2, found);
    if (!found) return;
    d = sqrt(best2);
  }
  if (d <= (1.0 + WIN_K) * rmax) atomicOr(bad, 1);
}

static int pyr_ready(mpg_grid_s *g, int st, hipStream_t s) {
  if (g->pts[st].n == 0 || g->pyr[st].built) return MPG_SUCCESS;
  return mpg_k_build_pyramid(g->pts[st], g->snx[st], g->sny[st], g->pyr[st], s);
}

// m: nCells / nVertices / maxEdges set, m->cell holding every cell centre.  Fills the geometry window.
int mpg_k_mesh_window(mpg_mesh_s *m, mpg_grid_s *g, const double *latVertex, const double *lonVertex, const int32_t *verticesOnCell, hipStream_t s) {
  int rc;
  // the grid as a set of boxes: the padded 4 x 4-cell boxes of its CORNER mesh when it has one (they cover the points of every
  // stagger, without gaps), else the 4 x 4-point boxes of the point staggers it has
  WinPyrs pyrs;
  pyrs.n = 0;
  if (g->pts[MPG_STAGGERLOC_CORNER].n == (int64_t)(g->nx + 1) * (g->ny + 1)) {
    if (!g->cellpyr.built && (rc = mpg_k_build_cell_pyramid(g->pts[MPG_STAGGERLOC_CORNER], g->nx, g->ny, g->cellpyr, s))) return rc;
    pyrs.v[pyrs.n++] = mpg_pyr_view(g->cellpyr);
  } else {
    for (int st : {MPG_STAGGERLOC_CENTER, MPG_STAGGERLOC_EDGE1, MPG_STAGGERLOC_EDGE2}) {
      if (g->pts[st].n == 0) continue;
      if ((rc = pyr_ready(g, st, s))) return rc;
      pyrs.v[pyrs.n++] = mpg_pyr_view(g->pyr[st]);
    }
  }
  const int64_t nC = m->nCells, nV = m->nVertices;
  const unsigned nbC = (unsigned)((nC + 255) / 256);
  TmpBuf<unsigned long long> stats;
  TmpBuf<float> D;
  if ((rc = stats.alloc(8, s)) || (rc = D.alloc((size_t)nC, s))) return rc;
  unsigned long long hs[8];
  // grid spacing h: the first margin is 6 h (a mesh as fine as the grid or finer); the first cut also counts the cells inside
  // the grid's boxes, and a mesh that turns out COARSER than the grid -- spacing ~ h * sqrt(points / cells inside) -- is cut again
  MPG_HIP(hipMemsetAsync(stats.p + 4, 0, sizeof(unsigned long long), s));
  {
    const PointSet &ctr = g->pts[MPG_STAGGERLOC_CENTER];
    k_grid_spacing<<<1, 64, 0, s>>>(g->nx, g->ny, ctr.x.p, ctr.y.p, ctr.z.p, stats.p + 4);
  }
  MPG_HIP(hipGetLastError());
  MPG_HIP(hipMemcpyAsync(hs + 4, stats.p + 4, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  double h_grid;
  memcpy(&h_grid, &hs[4], sizeof(double));
  if (!(h_grid > 0.0)) h_grid = 1e-4;   // a 1 x 1 grid
  const double P = (double)g->nx * (double)g->ny;
  double delta = 6.0 * h_grid;
  bool spacing_known = false;
  for (int attempt = 0;; ++attempt) {
    // (a mesh of a few dozen cells is taken whole: nothing to gain, and a grid inside ONE of its cells can be further than any
    // margin tried here from every cell centre -- it would look like a grid off the mesh)
    bool whole = attempt >= 6 || delta >= 2.0 || nC < 64;
    int64_t c0 = 0, c1 = nC;
    if (!whole) {
      MPG_HIP(hipMemsetAsync(stats.p, 0, 4 * sizeof(unsigned long long), s));
      MPG_HIP(hipMemsetAsync(stats.p + 2, 0xff, sizeof(unsigned long long), s));
      k_cell_dist<<<nbC, 256, 0, s>>>(nC, m->cell.x.p, m->cell.y.p, m->cell.z.p, pyrs, 2.0 * delta, delta, D.p, stats.p);
      MPG_HIP(hipGetLastError());
      MPG_HIP(hipMemcpyAsync(hs, stats.p, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
      MPG_HIP(hipStreamSynchronize(s));
      if (!spacing_known && hs[0] > 0) {   // cells per grid point inside the boxes -> the mesh's spacing there
        spacing_known = true;
        const double s_mesh = h_grid * sqrt(P / (double)hs[0]);
        if (6.0 * s_mesh > 1.5 * delta) {   // a mesh coarser than the grid: the margin follows the MESH's spacing
          delta = 6.0 * s_mesh;
          --attempt;
          continue;
        }
      }
      if (hs[1] == 0) {   // no cell within delta of the grid
        if (delta < 0.5) {                  // ... yet: a grid inside one large cell, or off the mesh -- look further, up to half a
          delta = fmin(4.0 * delta, 0.5);   // radian however small the first margin was (a row of a fine polar grid under a coarse
          --attempt;                        // mesh starts 400 times below the mesh's spacing), before concluding that the grid
          continue;                         // sees no cell; these steps do not use up the closure's attempts
        }
        c0 = c1 = 0;
      } else {
        c0 = (int64_t)hs[2];
        c1 = (int64_t)hs[3];
      }
      if (c1 - c0 > (nC * 9) / 10) whole = true;   // nothing to gain: take the mesh as it is, no closure question
    }
    if (whole) {
      c0 = 0;
      c1 = nC;
    }
    m->cw0 = c0;
    m->cwn = c1 - c0;
    m->voc.free();
    m->max_valence = -1;
    m->vert.free();
    m->tri.free();
    m->vw0 = m->vwn = 0;
    m->geo_margin = whole ? 4.0 : delta;
    if (m->cwn == 0) return MPG_SUCCESS;   // the grid sees no cell at all
    if ((rc = m->voc.alloc((size_t)m->cwn * m->maxEdges))) return rc;
    MPG_HIP(hipMemcpyAsync(m->voc.p, verticesOnCell + c0 * m->maxEdges, sizeof(int32_t) * (size_t)m->cwn * m->maxEdges, hipMemcpyHostToDevice, s));
    int64_t v0 = 0, v1 = nV;
    {   // the vertex numbers the rows reference: the window's vertex range, and a check of the table itself (a number beyond nVertices
        // would be read as a coordinate index by every geometry kernel)
      MPG_HIP(hipMemsetAsync(stats.p, 0xff, sizeof(unsigned long long), s));
      MPG_HIP(hipMemsetAsync(stats.p + 1, 0, sizeof(unsigned long long), s));
      k_vertex_range<<<(unsigned)std::min<int64_t>((m->cwn * m->maxEdges + 255) / 256, 4096), 256, 0, s>>>(m->cwn * m->maxEdges, m->voc.p, stats.p);
      MPG_HIP(hipGetLastError());
      MPG_HIP(hipMemcpyAsync(hs, stats.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
      MPG_HIP(hipStreamSynchronize(s));
      if ((int64_t)hs[1] > nV) {
        mpg_set_error("mpg_mesh_create_window: verticesOnCell refers to vertex %lld of %lld", (long long)hs[1], (long long)nV);
        return MPG_ERR_INVALID_ARG;
      }
      if (!whole) {
        if (hs[1] == 0) {
          v0 = v1 = 0;
        } else {
          v0 = (int64_t)hs[0];
          v1 = (int64_t)hs[1];
        }
      }
    }
    if (!whole && (v1 - v0) > (nV * 7) / 10) {   // numbering without bands (e.g. a Morton-ordered global mesh): the covering ranges are
      delta = 4.0;                               // most of the mesh, the vertex arrays -- the larger upload -- would hardly shrink:
      attempt = 6;                               // take the whole mesh (next pass of the loop)
      continue;
    }
    m->vw0 = v0;
    m->vwn = v1 - v0;
    if (m->vwn == 0) {   // rows of padding only
      MPG_HIP(hipStreamSynchronize(s));
      return MPG_SUCCESS;
    }
    if ((rc = m->vert.alloc(m->vwn))) return rc;
    {
      TmpBuf<double> tmp;
      if ((rc = tmp.alloc(2 * (size_t)m->vwn, s))) return rc;
      MPG_HIP(hipMemcpyAsync(tmp.p, lonVertex + v0, sizeof(double) * m->vwn, hipMemcpyHostToDevice, s));
      MPG_HIP(hipMemcpyAsync(tmp.p + m->vwn, latVertex + v0, sizeof(double) * m->vwn, hipMemcpyHostToDevice, s));
      TmpBuf<unsigned long long> cbad;
      if ((rc = cbad.alloc(1, s))) return rc;
      MPG_HIP(hipMemsetAsync(cbad.p, 0xff, sizeof(unsigned long long), s));
      if ((rc = mpg_k_mesh_coords_dev(m->vwn, tmp.p, tmp.p + m->vwn, m->vert.x.p, m->vert.y.p, m->vert.z.p, cbad.p, s))) return rc;
      unsigned long long hb = ~0ull;
      MPG_HIP(hipMemcpyAsync(&hb, cbad.p, sizeof(hb), hipMemcpyDeviceToHost, s));
      MPG_HIP(hipStreamSynchronize(s));
      if (hb != ~0ull) {
        mpg_set_error("mpg_mesh_create_window: vertex %lld has latitude %.17g, longitude %.17g -- not angles in RADIANS (|lat| <= pi/2, finite)",
                      (long long)(v0 + (int64_t)hb), latVertex[v0 + (int64_t)hb], lonVertex[v0 + (int64_t)hb]);
        return MPG_ERR_INVALID_ARG;
      }
    }   // tmp goes back to the pool in stream order; the host arrays are in use until the synchronisation that ends every path below
    if ((rc = m->tri.alloc(3 * (size_t)m->vwn))) return rc;
    TmpBuf<int32_t> cnt;
    if ((rc = cnt.alloc((size_t)m->vwn, s))) return rc;
    if ((rc = mpg_k_tri_scatter(m, cnt.p, s))) return rc;
    int32_t bad = 0;
    if (!whole) {
      TmpBuf<uint8_t> status;
      TmpBuf<int32_t> badd;
      if ((rc = status.alloc((size_t)m->vwn, s)) || (rc = badd.alloc(1, s))) return rc;
      MPG_HIP(hipMemsetAsync(badd.p, 0, sizeof(int32_t), s));
      k_win_vertex_status<<<(unsigned)((m->vwn + 255) / 256), 256, 0, s>>>(m->vwn, cnt.p, m->tri.p, m->cell.x.p, m->cell.y.p, m->cell.z.p, m->vert.x.p,
                                                                          m->vert.y.p, m->vert.z.p, D.p, delta, status.p);
      k_win_cell_closed<<<(unsigned)((m->cwn + 255) / 256), 256, 0, s>>>(m->cwn, m->cw0, m->vw0, m->maxEdges, m->voc.p, m->cell.x.p, m->cell.y.p,
                                                                        m->cell.z.p, m->vert.x.p, m->vert.y.p, m->vert.z.p, D.p, status.p, badd.p, pyrs,
                                                                        2.0 * delta);
      MPG_HIP(hipGetLastError());
      MPG_HIP(hipMemcpyAsync(&bad, badd.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
      MPG_HIP(hipStreamSynchronize(s));
    }
    if (bad) {   // a cell near the grid may be missing a neighbour that lies outside the window: cut wider
      delta *= 2.0;
      continue;
    }
    return mpg_k_tri_canon(m, cnt.p, s);
  }
}

// mpg_mesh_create's check of the table it was handed: a vertex number beyond nVertices would be read as a coordinate index by every
// geometry kernel (numbers <= 0 are padding, model_grid.F90:448)
int mpg_k_voc_check(const int32_t *voc_dev, int64_t nent, int64_t nV, const char *who, hipStream_t s) {
  if (nent <= 0) return MPG_SUCCESS;
  int rc;
  TmpBuf<unsigned long long> st;
  if ((rc = st.alloc(2, s))) return rc;
  MPG_HIP(hipMemsetAsync(st.p, 0xff, sizeof(unsigned long long), s));
  MPG_HIP(hipMemsetAsync(st.p + 1, 0, sizeof(unsigned long long), s));
  k_vertex_range<<<(unsigned)std::min<int64_t>((nent + 255) / 256, 4096), 256, 0, s>>>(nent, voc_dev, st.p);
  MPG_HIP(hipGetLastError());
  unsigned long long hs[2] = {0, 0};
  MPG_HIP(hipMemcpyAsync(hs, st.p, sizeof(hs), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  if ((int64_t)hs[1] > nV) {
    mpg_set_error("%s: verticesOnCell refers to vertex %lld of %lld", who, (long long)hs[1], (long long)nV);
    return MPG_ERR_INVALID_ARG;
  }
  return MPG_SUCCESS;
}

// mpg_init loads this translation unit's code object ahead of its first launch (mpg_api.hip: warm_modules)
const void *mpg_anchor_k_mesh_window() { return (const void *)k_vertex_range; }
