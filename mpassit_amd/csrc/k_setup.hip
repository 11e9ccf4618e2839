// Geometry ingest kernels: lon/lat -> unit sphere, dual (Delaunay) triangles from verticesOnCell,
// AABB pyramids over the structured target grid.
//
// Replaces the geometry half of ESMF_MeshCreate / ESMF_GridAddCoord as used by the reference
// (model_grid.F90:446-497 mesh definition: coordinate conversion :450-454,464-468, element corner
// count :448, connectivity :474-485; grid staggers :736-1038).  The reference's two quadratic host
// loops (unique_sort :2160-2178, FINDLOC :480) are unnecessary here: node ids are used directly.
// No floating-point contraction in this translation unit (see k_store_conserve.hip): what it computes -- weights, coordinates --
// is a function of the source text, not of which product the compiler chooses to fuse; explicit fma() calls stay what they are.
#pragma clang fp contract(off)
#include "geom.h"
#include "mpg_internal.h"

// ---- coordinates ---------------------------------------------------------------------------------
// mesh: radians in, degrees + wrap to (-180,180] exactly as model_grid.F90:450-454 (PI = 4*atan(1)),
// then degrees -> unit sphere (ESMF_COORDSYS_SPH_DEG).
__global__ __launch_bounds__(256) void k_mesh_coords(int64_t n, const double *__restrict__ lon_rad,
                                                     const double *__restrict__ lat_rad, double *__restrict__ x,
                                                     double *__restrict__ y, double *__restrict__ z, unsigned long long *__restrict__ bad) {
  const double PI = 3.14159265358979323846;  // == 4*atan(1) in float64
  const double d2r = 3.141592653589793 / 180.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    // a coordinate that is no angle (NaN / Inf, or a latitude beyond the poles: degrees handed over as radians) is reported, first index
    if (bad && (!(fabs(lat_rad[i]) <= 1.5707963267948966 + 1e-6) || !(fabs(lon_rad[i]) <= 1e3))) atomicMin(bad, (unsigned long long)i);
    double lo = lon_rad[i] * 180.0 / PI;
    if (lo > 180.0) lo -= 360.0;
    double la = lat_rad[i] * 180.0 / PI;
    lo *= d2r;
    la *= d2r;
    double sl, cl, so, co;
    sincos(la, &sl, &cl);
    sincos(lo, &so, &co);
    x[i] = cl * co;
    y[i] = cl * so;
    z[i] = sl;
  }
}
__global__ __launch_bounds__(256) void k_grid_coords(int64_t n, const double *__restrict__ lon_deg,
                                                     const double *__restrict__ lat_deg, double *__restrict__ x,
                                                     double *__restrict__ y, double *__restrict__ z, unsigned long long *__restrict__ bad) {
  const double d2r = 3.141592653589793 / 180.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    // (a corner row of a global lat-lon grid lies half a cell BEYOND the pole, e.g. 90.5: such values are angles and pass)
    if (bad && (!(fabs(lat_deg[i]) <= 180.0) || !(fabs(lon_deg[i]) <= 1e5))) atomicMin(bad, (unsigned long long)i);
    double lo = lon_deg[i] * d2r, la = lat_deg[i] * d2r;
    double sl, cl, so, co;
    sincos(la, &sl, &cl);
    sincos(lo, &so, &co);
    x[i] = cl * co;
    y[i] = cl * so;
    z[i] = sl;
  }
}

static inline int grid_for(int64_t n, int block = 256, int cap = 8192) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// lon/lat are HOST pointers: staged through a temporary device buffer
static int coords_common(bool mesh, int64_t n, const double *lon, const double *lat, PointSet &out, hipStream_t s) {
  int rc;
  if ((rc = out.alloc(n))) return rc;
  if (n == 0) return MPG_SUCCESS;
  TmpBuf<double> tmp;
  if ((rc = tmp.alloc(2 * (size_t)n, s))) return rc;
  TmpBuf<unsigned long long> bad;
  if ((rc = bad.alloc(1, s))) return rc;
  MPG_HIP(hipMemsetAsync(bad.p, 0xff, sizeof(unsigned long long), s));
  // pageable host arrays through the runtime's own staging: 45-50 GB/s once it is warm (a threaded pinned-buffer pipeline of
  // ours measured SLOWER: mpg_mesh_create of configuration 4 7.9-8.8 ms against 4.5, profiles/r04_first_call.txt)
  MPG_HIP(hipMemcpyAsync(tmp.p, lon, sizeof(double) * n, hipMemcpyHostToDevice, s));
  MPG_HIP(hipMemcpyAsync(tmp.p + n, lat, sizeof(double) * n, hipMemcpyHostToDevice, s));
  if (mesh)
    k_mesh_coords<<<grid_for(n), 256, 0, s>>>(n, tmp.p, tmp.p + n, out.x.p, out.y.p, out.z.p, bad.p);
  else
    k_grid_coords<<<grid_for(n), 256, 0, s>>>(n, tmp.p, tmp.p + n, out.x.p, out.y.p, out.z.p, bad.p);
  MPG_HIP(hipGetLastError());
  unsigned long long hb = ~0ull;
  MPG_HIP(hipMemcpyAsync(&hb, bad.p, sizeof(hb), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  tmp.free();
  if (hb != ~0ull) {
    mpg_set_error(mesh ? "mesh coordinates: point %lld has latitude %.17g, longitude %.17g -- not angles in RADIANS (|lat| <= pi/2, finite)"
                       : "grid coordinates: point %lld has latitude %.17g, longitude %.17g -- not angles in DEGREES (finite, |lat| <= 180)",
                  (long long)hb, lat[hb], lon[hb]);
    return MPG_ERR_INVALID_ARG;
  }
  return MPG_SUCCESS;
}
// bad_dev (may be NULL): one word preset to ~0, receives the first index whose coordinates are no angles
int mpg_k_mesh_coords_dev(int64_t n, const double *lon_rad_dev, const double *lat_rad_dev, double *x, double *y, double *z, unsigned long long *bad_dev,
                          hipStream_t s) {
  if (n > 0) k_mesh_coords<<<grid_for(n), 256, 0, s>>>(n, lon_rad_dev, lat_rad_dev, x, y, z, bad_dev);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}
int mpg_k_mesh_coords(int64_t n, const double *lon_rad, const double *lat_rad, PointSet &out, hipStream_t s) {
  return coords_common(true, n, lon_rad, lat_rad, out, s);
}
int mpg_k_grid_coords(int64_t n, const double *lon_deg, const double *lat_deg, PointSet &out, hipStream_t s) {
  return coords_common(false, n, lon_deg, lat_deg, out, s);
}

// ---- dual triangles ---------------------------------------------------------------
// pass 1: every (cell, vertex) incidence claims a slot of its vertex with an atomic counter
// voc: the rows of cells cell0 .. cell0 + nCells - 1; cnt / tri: the vertices vert0 .. vert0 + nVertices - 1 (the mesh's
// geometry window: everything for mpg_mesh_create); tri holds GLOBAL cell ids
__global__ __launch_bounds__(256) void k_tri_scatter(int64_t nCells, int maxEdges, int64_t nVertices,
                                                     const int32_t *__restrict__ voc, int32_t *__restrict__ cnt,
                                                     int32_t *__restrict__ tri /*[3][nV]*/, int64_t cell0, int64_t vert0) {
  int64_t total = nCells * maxEdges;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t v = voc[e];
    if (v <= 0) continue;  // 0 = padding (model_grid.F90:448)
    v -= 1 + vert0;
    if (v < 0 || v >= nVertices) continue;
    int slot = atomicAdd(&cnt[v], 1);
    if (slot < 3) tri[(int64_t)slot * nVertices + v] = (int32_t)(cell0 + e / maxEdges);
  }
}
// pass 2: canonical order (ascending ids, then CCW) so the result is independent of atomic ordering
__global__ __launch_bounds__(256) void k_tri_canon(int64_t nVertices, const int32_t *__restrict__ cnt,
                                                   int32_t *__restrict__ tri, const double *__restrict__ cx,
                                                   const double *__restrict__ cy, const double *__restrict__ cz,
                                                   unsigned long long *__restrict__ nvalid) {
  __shared__ int swave[4];
  int mine = 0;
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nVertices; v += (int64_t)gridDim.x * blockDim.x) {
    bool ok = false;
    int32_t a = tri[v], b = tri[nVertices + v], c = tri[2 * nVertices + v];
    if (cnt[v] == 3) {
      int32_t t;
      if (a > b) { t = a; a = b; b = t; }
      if (b > c) { t = b; b = c; c = t; }
      if (a > b) { t = a; a = b; b = t; }
      dv3 A = ld3(cx, cy, cz, a), B = ld3(cx, cy, cz, b), C = ld3(cx, cy, cz, c);
      double d = det3_from(A, B, C);
      if (d < 0.0) { t = b; b = c; c = t; }
      ok = d != 0.0;
    }
    if (!ok) a = b = c = -1;
    tri[v] = a;
    tri[nVertices + v] = b;
    tri[2 * nVertices + v] = c;
    mine += ok;
  }
  // one atomic per workgroup (a few thousand in all): per-wave atomics on the one counter queued behind each other
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
  if ((threadIdx.x & 63) == 0) swave[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = swave[0] + swave[1] + swave[2] + swave[3];
    if (tot) atomicAdd(nvalid, (unsigned long long)tot);
  }
}

// Two steps so that mpg_mesh_create_window can look at the raw incidence counts in between (k_mesh_window.hip: which
// vertices of the window miss a cell that exists outside it): scatter leaves cnt[v] = number of cells of the window's rows
// touching vertex v and their ids in tri's slots; canon turns complete vertices into canonical triangles and clears the rest.
int mpg_k_tri_scatter(mpg_mesh_s *m, int32_t *cnt, hipStream_t s) {
  const int64_t nV = m->vwn;
  if (nV == 0) return MPG_SUCCESS;
  MPG_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * nV, s));
  MPG_HIP(hipMemsetAsync(m->tri.p, 0xff, sizeof(int32_t) * 3 * nV, s));
  if (m->cwn > 0)
    k_tri_scatter<<<grid_for(m->cwn * m->maxEdges), 256, 0, s>>>(m->cwn, m->maxEdges, nV, m->voc.p, cnt, m->tri.p, m->cw0, m->vw0);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}
int mpg_k_tri_canon(mpg_mesh_s *m, const int32_t *cnt, hipStream_t s) {
  int rc;
  const int64_t nV = m->vwn;
  m->nTriValid = 0;
  if (nV == 0) return MPG_SUCCESS;
  TmpBuf<unsigned long long> nv;
  if ((rc = nv.alloc(1, s))) return rc;
  MPG_HIP(hipMemsetAsync(nv.p, 0, sizeof(unsigned long long), s));
  int64_t nb = (nV + 255) / 256;
  if (nb > 4096) nb = 4096;   // grid-stride: 16 waves per CU in flight, a handful of vertices per thread
  k_tri_canon<<<(unsigned)nb, 256, 0, s>>>(nV, cnt, m->tri.p, m->cell.x.p, m->cell.y.p, m->cell.z.p, nv.p);
  MPG_HIP(hipGetLastError());
  unsigned long long h = 0;
  MPG_HIP(hipMemcpyAsync(&h, nv.p, sizeof(h), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  m->nTriValid = (int64_t)h;
  return MPG_SUCCESS;
}
int mpg_k_dual_triangles(mpg_mesh_s *m, hipStream_t s) {
  int rc;
  m->nTriValid = 0;
  if (m->vwn == 0) return MPG_SUCCESS;
  if ((rc = m->tri.alloc(3 * (size_t)m->vwn))) return rc;
  TmpBuf<int32_t> cnt;
  if ((rc = cnt.alloc((size_t)m->vwn, s))) return rc;
  if ((rc = mpg_k_tri_scatter(m, cnt.p, s))) return rc;
  return mpg_k_tri_canon(m, cnt.p, s);
}

// ---- AABB pyramid over a structured point set --------------------------------------------------------
// level 0: one node per B0 x B0 block of points.  `halo` = 1 builds cell boxes instead: node covers
// cells [i0,i0+B0) x [j0,j0+B0) whose corners are points [i0..i0+B0] x [j0..j0+B0] of an (nx+1)x(ny+1) set.
__global__ __launch_bounds__(256) void k_pyr_leaf(int npx, int npy, int nbx, int nby, int halo,
                                                  const double *__restrict__ x, const double *__restrict__ y,
                                                  const double *__restrict__ z, double *__restrict__ box) {
  int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= (int64_t)nbx * nby) return;
  int bx = (int)(b % nbx), by = (int)(b / nbx);
  int i0 = bx * MPG_PYR_B0, j0 = by * MPG_PYR_B0;
  int i1 = min(i0 + MPG_PYR_B0 + halo, npx), j1 = min(j0 + MPG_PYR_B0 + halo, npy);
  double lo[3] = {2, 2, 2}, hi[3] = {-2, -2, -2};
  for (int j = j0; j < j1; ++j)
    for (int i = i0; i < i1; ++i) {
      int64_t p = (int64_t)j * npx + i;
      double v[3] = {x[p], y[p], z[p]};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        lo[k] = fmin(lo[k], v[k]);
        hi[k] = fmax(hi[k], v[k]);
      }
    }
  if (halo) {
    // cell boxes: the spherical quads bulge beyond the planar hull of their corners by <= diameter^2/2
    double d2 = (hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]);
    double pad = 0.5 * d2 + 1e-9;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      lo[k] -= pad;
      hi[k] += pad;
    }
  }
  double *o = box + 6 * b;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2];
  o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2];
}
__global__ __launch_bounds__(256) void k_pyr_up(int cnx, int cny, int pnx, int pny, const double *__restrict__ child,
                                                double *__restrict__ parent) {
  int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= (int64_t)pnx * pny) return;
  int bx = (int)(b % pnx), by = (int)(b / pnx);
  double lo[3] = {2, 2, 2}, hi[3] = {-2, -2, -2};
  for (int dj = 0; dj < 2; ++dj)
    for (int di = 0; di < 2; ++di) {
      int ci = 2 * bx + di, cj = 2 * by + dj;
      if (ci >= cnx || cj >= cny) continue;
      const double *c = child + 6 * ((int64_t)cj * cnx + ci);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        lo[k] = fmin(lo[k], c[k]);
        hi[k] = fmax(hi[k], c[3 + k]);
      }
    }
  double *o = parent + 6 * b;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2];
  o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2];
}

static int build_pyr(const PointSet &pts, int npx, int npy, int halo, Pyramid &pyr, hipStream_t s) {
  // node grid of level 0 counts blocks of points (halo=0) or of cells (halo=1: npx-1 x npy-1 cells)
  int ux = npx - halo, uy = npy - halo;
  if (ux < 1 || uy < 1) {
    mpg_set_error("pyramid: empty grid");
    return MPG_ERR_INVALID_ARG;
  }
  int lx = (ux + MPG_PYR_B0 - 1) / MPG_PYR_B0, ly = (uy + MPG_PYR_B0 - 1) / MPG_PYR_B0;
  int nlev = 0;
  int64_t total = 0;
  while (true) {
    if (nlev >= MPG_PYR_MAXLEV) {
      mpg_set_error("pyramid: too many levels");
      return MPG_ERR_OVERFLOW;
    }
    pyr.nx[nlev] = lx;
    pyr.ny[nlev] = ly;
    pyr.off[nlev] = total;
    total += (int64_t)lx * ly;
    ++nlev;
    if (lx == 1 && ly == 1) break;
    lx = (lx + 1) / 2;
    ly = (ly + 1) / 2;
  }
  pyr.off[nlev] = total;
  pyr.nlev = nlev;
  int rc;
  if ((rc = pyr.box.alloc(6 * (size_t)total))) return rc;
  int64_t n0 = (int64_t)pyr.nx[0] * pyr.ny[0];
  k_pyr_leaf<<<(unsigned)((n0 + 255) / 256), 256, 0, s>>>(npx, npy, pyr.nx[0], pyr.ny[0], halo, pts.x.p, pts.y.p, pts.z.p, pyr.box.p);
  for (int l = 1; l < nlev; ++l) {
    int64_t nl = (int64_t)pyr.nx[l] * pyr.ny[l];
    k_pyr_up<<<(unsigned)((nl + 255) / 256), 256, 0, s>>>(pyr.nx[l - 1], pyr.ny[l - 1], pyr.nx[l], pyr.ny[l],
                                                         pyr.box.p + 6 * pyr.off[l - 1], pyr.box.p + 6 * pyr.off[l]);
  }
  MPG_HIP(hipGetLastError());
  pyr.built = true;
  return MPG_SUCCESS;
}
int mpg_k_build_pyramid(const PointSet &pts, int nx, int ny, Pyramid &pyr, hipStream_t s) {
  return build_pyr(pts, nx, ny, 0, pyr, s);
}
int mpg_k_build_cell_pyramid(const PointSet &corner, int nx, int ny, Pyramid &pyr, hipStream_t s) {
  return build_pyr(corner, nx + 1, ny + 1, 1, pyr, s);
}
PyramidView mpg_pyr_view(const Pyramid &p) {
  PyramidView v;
  v.nlev = p.nlev;
  for (int i = 0; i < MPG_PYR_MAXLEV; ++i) {
    v.nx[i] = p.nx[i];
    v.ny[i] = p.ny[i];
  }
  for (int i = 0; i <= MPG_PYR_MAXLEV; ++i) v.off[i] = p.off[i];
  v.box = p.box.p;
  return v;
}

// mpg_init loads this translation unit's code object ahead of its first launch (mpg_api.hip: warm_modules)
const void *mpg_anchor_k_setup() { return (const void *)k_mesh_coords; }
