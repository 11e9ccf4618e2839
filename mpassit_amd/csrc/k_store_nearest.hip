// K3 "build_nearest": nearest source-to-destination RegridStore.
//
// Replaces ESMF_FieldBundleRegridStore(regridmethod=NEAREST_STOD) at interp.F90:421 (and the soil
// bundle at :437 by method fall-through).  Semantics (App. A6): every destination
// point maps to argmin over ALL source cell centres of the 3-D chord distance, ties -> lowest cell
// id; points outside the mesh footprint are mapped too.
//
// MI355X-native formulation: cell centres are sorted along a 63-bit Morton curve (rocPRIM radix sort,
// set-up only), which makes an implicit 8-ary BVH: leaf = 8 consecutive sorted sites, each upper level
// groups 8 nodes.  One thread per target point runs an exact branch-and-bound descent (nearest child
// first).  Distances and box lower bounds are evaluated without FMA in a fixed order so the bound is
// monotone in floating point and the argmin / tie-break is identical to the CPU oracle's.
#include <cstring>


#include <string.h>

#include <algorithm>

#include "geom.h"
#include "mpg_internal.h"

__device__ __forceinline__ unsigned long long spread21(unsigned long long v) {
  v &= 0x1fffffull;
  v = (v | v << 32) & 0x1f00000000ffffull;
  v = (v | v << 16) & 0x1f0000ff0000ffull;
  v = (v | v << 8) & 0x100f00f00f00f00full;
  v = (v | v << 4) & 0x10c30c30c30c30c3ull;
  v = (v | v << 2) & 0x1249249249249249ull;
  return v;
}
__global__ __launch_bounds__(256) void k_morton(int64_t n, const double *__restrict__ x, const double *__restrict__ y,
                                                const double *__restrict__ z, unsigned long long *__restrict__ key,
                                                int32_t *__restrict__ id, int64_t first) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  x += first; y += first; z += first;   // sites = cells first .. first + n - 1 (the mesh's geometry window)
  const double sc = 2097151.0 * 0.5;  // (2^21 - 1) / 2
  unsigned long long qx = (unsigned long long)fmin(fmax((x[i] + 1.0) * sc, 0.0), 2097151.0);
  unsigned long long qy = (unsigned long long)fmin(fmax((y[i] + 1.0) * sc, 0.0), 2097151.0);
  unsigned long long qz = (unsigned long long)fmin(fmax((z[i] + 1.0) * sc, 0.0), 2097151.0);
  key[i] = spread21(qx) | (spread21(qy) << 1) | (spread21(qz) << 2);
  id[i] = (int32_t)(first + i);
}
__global__ __launch_bounds__(256) void k_gather_sites(int64_t n, const int32_t *__restrict__ id, const double *__restrict__ x,
                                                      const double *__restrict__ y, const double *__restrict__ z,
                                                      double *__restrict__ sx, double *__restrict__ sy, double *__restrict__ sz) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t c = id[i];
  sx[i] = x[c];
  sy[i] = y[c];
  sz[i] = z[c];
}
__global__ __launch_bounds__(256) void k_bvh_leaf(int64_t n, int64_t nleaf, const double *__restrict__ sx,
                                                  const double *__restrict__ sy, const double *__restrict__ sz,
                                                  double *__restrict__ box) {
  int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= nleaf) return;
  double lo[3] = {2, 2, 2}, hi[3] = {-2, -2, -2};
  int64_t e = min(n, (b + 1) * MPG_BVH_LEAF);
  for (int64_t i = b * MPG_BVH_LEAF; i < e; ++i) {
    lo[0] = fmin(lo[0], sx[i]); hi[0] = fmax(hi[0], sx[i]);
    lo[1] = fmin(lo[1], sy[i]); hi[1] = fmax(hi[1], sy[i]);
    lo[2] = fmin(lo[2], sz[i]); hi[2] = fmax(hi[2], sz[i]);
  }
  double *o = box + 6 * b;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2];
}
__global__ __launch_bounds__(256) void k_bvh_up(int64_t nchild, int64_t nparent, const double *__restrict__ child,
                                                double *__restrict__ parent) {
  int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= nparent) return;
  double lo[3] = {2, 2, 2}, hi[3] = {-2, -2, -2};
  int64_t e = min(nchild, (b + 1) * MPG_BVH_FAN);
  for (int64_t i = b * MPG_BVH_FAN; i < e; ++i) {
    const double *c = child + 6 * i;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      lo[k] = fmin(lo[k], c[k]);
      hi[k] = fmax(hi[k], c[3 + k]);
    }
  }
  double *o = parent + 6 * b;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2];
}

// whole = false: only the cells of the mesh's geometry window are sites (a mesh cut to one rank's grid,
// mpg_mesh_create_window); every cell centre is on the device in either case, so a search that turns out to need the
// cells outside the window (mpg_k_store_nearest checks) rebuilds with whole = true
int mpg_k_build_bvh(mpg_mesh_s *m, hipStream_t s, bool whole) {
  SiteBvh &b = m->bvh;
  if (b.built && (m->bvh_whole || !whole)) return MPG_SUCCESS;
  if (b.built) b.free();
  int rc;
  const int64_t first = whole ? 0 : m->cw0;
  int64_t n = whole ? m->nCells : m->cwn;
  if (n == 0) {   // an empty window: the whole mesh it is
    whole = true;
    n = m->nCells;
  }
  m->bvh_whole = whole;
  b.n = n;
  TmpBuf<unsigned long long> key_in, key_out;
  TmpBuf<int32_t> id_in;
  if ((rc = key_in.alloc(n, s)) || (rc = key_out.alloc(n, s)) || (rc = id_in.alloc(n, s)) || (rc = b.sorted_id.alloc(n))) return rc;
  if ((rc = b.sorted.alloc(n))) return rc;
  unsigned nb = (unsigned)((n + 255) / 256);
  k_morton<<<nb, 256, 0, s>>>(n, m->cell.x.p, m->cell.y.p, m->cell.z.p, key_in.p, id_in.p, first);
  if ((rc = mpg_sort_pairs_u64_i32(key_in.p, key_out.p, id_in.p, b.sorted_id.p, n, s))) return rc;   // (k_sort.hip: rocPRIM's radix sort)
  k_gather_sites<<<nb, 256, 0, s>>>(n, b.sorted_id.p, m->cell.x.p, m->cell.y.p, m->cell.z.p, b.sorted.x.p, b.sorted.y.p, b.sorted.z.p);
  // level sizes
  int nlev = 0;
  int64_t total = 0, cnt = (n + MPG_BVH_LEAF - 1) / MPG_BVH_LEAF;
  while (true) {
    if (nlev >= MPG_BVH_MAXLEV) {
      mpg_set_error("bvh: too many levels");
      return MPG_ERR_OVERFLOW;
    }
    b.nnodes[nlev] = cnt;
    b.off[nlev] = total;
    total += cnt;
    ++nlev;
    if (cnt == 1) break;
    cnt = (cnt + MPG_BVH_FAN - 1) / MPG_BVH_FAN;
  }
  b.off[nlev] = total;
  b.nlev = nlev;
  if ((rc = b.box.alloc(6 * (size_t)total))) return rc;
  k_bvh_leaf<<<(unsigned)((b.nnodes[0] + 255) / 256), 256, 0, s>>>(n, b.nnodes[0], b.sorted.x.p, b.sorted.y.p, b.sorted.z.p, b.box.p);
  for (int l = 1; l < nlev; ++l)
    k_bvh_up<<<(unsigned)((b.nnodes[l] + 255) / 256), 256, 0, s>>>(b.nnodes[l - 1], b.nnodes[l], b.box.p + 6 * b.off[l - 1],
                                                                  b.box.p + 6 * b.off[l]);
  MPG_HIP(hipGetLastError());
  MPG_HIP(hipStreamSynchronize(s));
  key_in.free(); key_out.free(); id_in.free();
  b.built = true;
  return MPG_SUCCESS;
}

static int g_nn_variant = 1;   // "nn_variant": 1 = wave-cooperative search (default), 0 = one thread per point
int mpg_nearest_variant() { return g_nn_variant; }
void mpg_set_nearest_variant(int v) { g_nn_variant = v; }

#define NN_STACK 96
// masked != 0: only the points whose out[] entry is negative are searched (the ones the index-space search could not settle)
__global__ __launch_bounds__(256) void k_nearest_query(int64_t P, const double *__restrict__ px, const double *__restrict__ py,
                                                       const double *__restrict__ pz, SiteBvhView b, int32_t *__restrict__ out, int masked) {
  int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p >= P) return;
  if (masked && out[p] >= 0) return;
  double X = px[p], Y = py[p], Z = pz[p];
  double best = INFINITY;
  int32_t best_id = 0x7fffffff;
  int stack[NN_STACK];
  int sp = 0;
  stack[sp++] = (b.nlev - 1) << 27;  // node < 2^27 per level
  while (sp > 0) {
    int e = stack[--sp];
    int lev = e >> 27;
    int64_t node = e & ((1 << 27) - 1);
    double dbox = boxdist2_nofma(X, Y, Z, b.box + 6 * (b.off[lev] + node));
    if (dbox > best) continue;  // equal bounds are explored: a tie with a lower id may hide inside
    if (lev == 0) {
      int64_t e1 = min(b.n, (node + 1) * MPG_BVH_LEAF);
      for (int64_t i = node * MPG_BVH_LEAF; i < e1; ++i) {
        double d = dist2_nofma(X, Y, Z, b.sx[i], b.sy[i], b.sz[i]);
        int32_t id = b.sid[i];
        if (d < best || (d == best && id < best_id)) {
          best = d;
          best_id = id;
        }
      }
    } else {
      // children sorted by decreasing bound so that the nearest is popped first
      int64_t c0 = node * MPG_BVH_FAN, c1 = min(b.nnodes[lev - 1], c0 + MPG_BVH_FAN);
      double cd[MPG_BVH_FAN];
      int ci[MPG_BVH_FAN];
      int nc = 0;
      for (int64_t c = c0; c < c1; ++c) {
        double d = boxdist2_nofma(X, Y, Z, b.box + 6 * (b.off[lev - 1] + c));
        if (d > best) continue;
        int k = nc++;
        while (k > 0 && cd[k - 1] < d) {
          cd[k] = cd[k - 1];
          ci[k] = ci[k - 1];
          --k;
        }
        cd[k] = d;
        ci[k] = (int)(c - c0);
      }
      for (int k = 0; k < nc && sp < NN_STACK; ++k) stack[sp++] = ((lev - 1) << 27) | (int)(c0 + ci[k]);
    }
  }
  out[p] = best_id;
}

// Wave-cooperative form of the same exact search.  A wavefront owns a patch of 8 x 8 neighbouring target points and walks
// the BVH ONCE for all of them: one shared stack (LDS), every node box and every leaf site fetched with wave-uniform
// (scalar) loads, each lane keeping its own (best, best_id) with the very comparisons of k_nearest_query.  A node is
// opened when ANY lane's bound admits it (wavefront ballot); a lane whose bound does not admit it evaluates it anyway,
// which cannot change its answer (every site in the box is at least as far as the box).  So the result is identical
// to the one-thread-per-point search, but the per-lane private stacks, the divergent descents and the 64-fold
// re-fetching of the same boxes are gone: neighbouring points share almost their whole search path.
// Children are pushed so that the one nearest to the patch (smallest bound of the patch's first point) is popped first.
#define NNW_STACK 128
#define NNW_WAVES 4
__global__ __launch_bounds__(64 * NNW_WAVES) void k_nearest_query_w(int npx, int npy, const double *__restrict__ px,
                                                                   const double *__restrict__ py, const double *__restrict__ pz,
                                                                   SiteBvhView b, int32_t *__restrict__ out, int32_t *__restrict__ overflow, int masked) {
  __shared__ int stk[NNW_WAVES][NNW_STACK];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nbx = (npx + 7) >> 3;
  const int64_t patch = (int64_t)blockIdx.x * NNW_WAVES + wave;
  const int64_t npatch = (int64_t)nbx * ((npy + 7) >> 3);
  if (patch >= npatch) return;
  const int i = (int)(patch % nbx) * 8 + (lane & 7), j = (int)(patch / nbx) * 8 + (lane >> 3);
  bool act = i < npx && j < npy;
  const int64_t p = act ? (int64_t)j * npx + i : 0;
  if (masked) act = act && out[p] < 0;      // settled already by the index-space search: this lane only rides along
  if (__ballot(act) == 0) return;
  const double X = px[p], Y = py[p], Z = pz[p];
  double best = act ? INFINITY : -1.0;      // an inactive lane admits nothing
  int32_t best_id = 0x7fffffff;
  // Seed: every lane first descends by itself to the leaf whose boxes are nearest to ITS point (neighbouring lanes take the
  // same path: the loads coalesce to a few lines) ...  A seed is a real site with its real (distance, id), so the exact search
  // that follows returns the same lexicographic minimum; what it buys is the bound the shared walk starts from (without any
  // seed the walk opened ~190 nodes per patch, because the bounds of the lanes far from the first leaf stayed wide).
  int64_t seed_leaf = -1;
  if (act) {
    int64_t node = 0;
    for (int lev = b.nlev - 1; lev > 0; --lev) {
      const int64_t c0 = node * MPG_BVH_FAN, c1 = min(b.nnodes[lev - 1], c0 + MPG_BVH_FAN);
      double dmin = INFINITY;
      int64_t cmin = c0;
      for (int64_t c = c0; c < c1; ++c) {
        const double d = boxdist2_nofma(X, Y, Z, b.box + 6 * (b.off[lev - 1] + c));
        if (d < dmin) {
          dmin = d;
          cmin = c;
        }
      }
      node = cmin;
    }
    seed_leaf = node;
  }
  // ... and every lane then tries the sites of ALL the distinct leaves the patch's lanes arrived at (about ten of them, each
  // fetched once with wave-uniform loads): a lane's true nearest site is almost always in its own leaf or a neighbour's, so
  // the shared walk starts from bounds that are exact for nearly every lane and only has to confirm them.
  for (unsigned long long todo = __ballot(seed_leaf >= 0); todo;) {
    const int leader = __ffsll((long long)todo) - 1;
    const int64_t leaf = ((int64_t)__builtin_amdgcn_readlane((int)(seed_leaf >> 32), leader) << 32) |
                         (unsigned)__builtin_amdgcn_readlane((int)seed_leaf, leader);
    const int64_t e1 = min(b.n, (leaf + 1) * MPG_BVH_LEAF);
    for (int64_t q = leaf * MPG_BVH_LEAF; q < e1; ++q) {
      const double d = dist2_nofma(X, Y, Z, b.sx[q], b.sy[q], b.sz[q]);
      const int32_t id = b.sid[q];
      if (d < best || (d == best && id < best_id)) {   // an inactive lane has best = -1: nothing is below it
        best = d;
        best_id = id;
      }
    }
    todo &= ~__ballot(seed_leaf == leaf);
  }
  int *st = stk[wave];
  int sp = 0;
  st[sp++] = (b.nlev - 1) << 27;
  while (sp > 0) {
    const int e = __builtin_amdgcn_readfirstlane(st[--sp]);
    const int lev = e >> 27;
    const int64_t node = e & ((1 << 27) - 1);
    const double dbox = boxdist2_nofma(X, Y, Z, b.box + 6 * (b.off[lev] + node));
    if (__ballot(!(dbox > best)) == 0) continue;     // equal bounds are explored: a tie with a lower id may hide inside
    if (lev == 0) {
      const int64_t e1 = min(b.n, (node + 1) * MPG_BVH_LEAF);
      for (int64_t q = node * MPG_BVH_LEAF; q < e1; ++q) {
        const double d = dist2_nofma(X, Y, Z, b.sx[q], b.sy[q], b.sz[q]);
        const int32_t id = b.sid[q];
        if (d < best || (d == best && id < best_id)) {
          best = d;
          best_id = id;
        }
      }
    } else {
      const int64_t c0 = node * MPG_BVH_FAN, c1 = min(b.nnodes[lev - 1], c0 + MPG_BVH_FAN);
      double key[MPG_BVH_FAN];   // ordering key: the bound seen by the patch's first lane (uniform)
      int ci[MPG_BVH_FAN];
      int nc = 0;
      for (int64_t c = c0; c < c1; ++c) {
        const double d = boxdist2_nofma(X, Y, Z, b.box + 6 * (b.off[lev - 1] + c));
        if (__ballot(!(d > best)) == 0) continue;
        const double k0 = __builtin_bit_cast(double, ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(__builtin_bit_cast(unsigned long long, d) >> 32)) << 32) |
                                                         (unsigned)__builtin_amdgcn_readfirstlane((int)__builtin_bit_cast(unsigned long long, d)));
        int k = nc++;
        while (k > 0 && key[k - 1] < k0) {
          key[k] = key[k - 1];
          ci[k] = ci[k - 1];
          --k;
        }
        key[k] = k0;
        ci[k] = (int)(c - c0);
      }
      if (sp + nc > NNW_STACK) {   // cannot happen (7 pushes per level, <= 12 levels); reported, never silently dropped
        if (lane == 0) atomicOr(overflow, 1);
        nc = NNW_STACK - sp;
      }
      for (int k = 0; k < nc; ++k) st[sp++] = ((lev - 1) << 27) | (int)(c0 + ci[k]);
    }
  }
  if (act) out[p] = best_id;
}

// largest squared distance between a target point and the site found for it (one atomic per workgroup; distances are
// non-negative doubles, whose bit patterns order like the values)
__global__ __launch_bounds__(256) void k_nn_max_d2(int64_t P, const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz,
                                                   const int32_t *__restrict__ idx, const double *__restrict__ cx, const double *__restrict__ cy,
                                                   const double *__restrict__ cz, unsigned long long *__restrict__ out) {
  __shared__ double sw[4];
  double mx = 0.0;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
    const int32_t c = idx[p];
    mx = fmax(mx, dist2_nofma(px[p], py[p], pz[p], cx[c], cy[c], cz[c]));
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o));
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out, (unsigned long long)__double_as_longlong(fmax(fmax(sw[0], sw[1]), fmax(sw[2], sw[3]))));
}

static int nearest_search(mpg_mesh_s *m, int npx, int npy, const PointSet &pts, mpg_handle_s *h, hipStream_t s, int masked = 0) {
  int rc;
  const int64_t P = (int64_t)npx * npy;
  SiteBvh &b = m->bvh;
  if (b.nnodes[0] >= (1 << 27)) {
    mpg_set_error("mesh too large for the nearest-neighbour BVH");
    return MPG_ERR_OVERFLOW;
  }
  SiteBvhView v;
  v.n = b.n;
  v.sx = b.sorted.x.p; v.sy = b.sorted.y.p; v.sz = b.sorted.z.p;
  v.sid = b.sorted_id.p;
  v.nlev = b.nlev;
  for (int i = 0; i < MPG_BVH_MAXLEV; ++i) v.nnodes[i] = b.nnodes[i];
  for (int i = 0; i <= MPG_BVH_MAXLEV; ++i) v.off[i] = b.off[i];
  v.box = b.box.p;
  if (mpg_nearest_variant() == 0) {   // "nn_variant" knob 0: the one-thread-per-point search (kept as the cross-check of the tests)
    k_nearest_query<<<(unsigned)((P + 255) / 256), 256, 0, s>>>(P, pts.x.p, pts.y.p, pts.z.p, v, h->idx.p, masked);
    MPG_HIP(hipGetLastError());
    MPG_HIP(hipStreamSynchronize(s));
    return MPG_SUCCESS;
  }
  TmpBuf<int32_t> ovf;
  if ((rc = ovf.alloc(1, s))) return rc;
  MPG_HIP(hipMemsetAsync(ovf.p, 0, sizeof(int32_t), s));
  const int64_t npatch = (int64_t)((npx + 7) / 8) * ((npy + 7) / 8);
  k_nearest_query_w<<<(unsigned)((npatch + NNW_WAVES - 1) / NNW_WAVES), 64 * NNW_WAVES, 0, s>>>(npx, npy, pts.x.p, pts.y.p, pts.z.p, v, h->idx.p,
                                                                                                ovf.p, masked);
  MPG_HIP(hipGetLastError());
  int32_t h_ovf = 0;
  MPG_HIP(hipMemcpyAsync(&h_ovf, ovf.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  if (h_ovf) {
    mpg_set_error("RegridStore(nearest): traversal stack of the wave-cooperative search overflowed");
    return MPG_ERR_OVERFLOW;
  }
  return MPG_SUCCESS;
}

// ---- nearest cell through the grid's own index space (round 4) --------------------------------------------------------------
// On a grid that knows its projection every cell centre has a place (i, j) in the grid's index space (k_points_ij,
// k_target_grid.hip); binned there, the cells around a target point are the contents of the few bins around the point's OWN
// index -- no tree, no sort.  Exactness does not rest on the map being isometric, only on a lower bound h of the chord length
// of one index unit (mpg_grid_min_index_chord: the map factor's maximum over the grid's latitudes and a margin): after the
// bins within r rings have been looked at, every cell not looked at is at least r * bin index units away, i.e. at least
// 0.8 * r * bin * h on the sphere, so a best distance below that is final -- ties included (an equally near cell is inside
// the bound and was compared: lowest id wins, with the distance arithmetic of the BVH search and the oracle).  A point that
// finds nothing final within NB_RINGS rings (the grid sticks far out of the mesh, the polar rows of a global grid) is left at
// -1 and the BVH search above finishes exactly those points (masked): never a different answer, only a slower one.  Cells
// whose index is not usable (NaN: the projection's pole / far side, the last degree before the poles on a lat-lon grid) are
// not binned, and no answer as far away as such a cell could be -- or as far away as the Lambert cut, across which index
// distance says nothing about distance on the sphere -- is called final (`cap` in k_nb_query).  Lat-lon grids take the bound h
// per point, global ones wrap their bins in i, regional ones unwrap the cells' indices about the grid's middle column.
#define NB_BIN_MIN 2    // index units (grid points) per bin side: 2 when the mesh is as fine as the grid, up to NB_BIN_MAX when it is
#define NB_BIN_MAX 16   // coarser (about one cell per bin: a 30-km mesh under a 3-km grid takes 9) -- the rings then reach the next cells
#define NB_RINGS 4      // rings of bins a point may look at; the bin grid extends that far beyond the grid's points
struct NbParams {
  int bin;             // index units per bin side
  int nbx, nby;        // bins
  int per, nxp;        // periodic in i (a global lat-lon grid): the index wraps at nxp, no margin in i
  float di, dj;        // stagger offsets of the point indices
  double h;            // chord length of one index unit, lower bound, times the safety factor (non-local form)
  double cap;          // nothing beyond this chord distance is final (cut meshes: the window's margin)
  int local;           // lat-lon: the bound is taken per point, h = 0.8 * min(dlat, dlon * cos(|lat| + what the rings span))
  double dlat, dlon;   // radians per index unit (lat-lon)
  int zone;            // 0 none; 1 lat-lon: no usable index beyond +-zlim degrees; 2 Lambert: beyond hemi * lat >= 89 or <= -60, and
  double zlim, hemi;   //   the index jumps across the cut meridian stdlon + 180; 3 polar stereographic: beyond hemi * lat <= -60 only;
  double stdlon;       //   4 Mercator: beyond +-zlim degrees and across the cut opposite the grid's middle column (stdlon = its longitude)
};
__device__ __forceinline__ bool nb_bin_of(const NbParams &q, float ci, float cj, int *bx, int *by) {
  float u = ci + q.di, v = cj + q.dj + (float)(NB_RINGS * q.bin);
  if (!(u == u) || !(v == v)) return false;
  if (q.per) {
    if (u < 0.f) u += (float)q.nxp;
    if (u >= (float)q.nxp) u -= (float)q.nxp;
  } else {
    u += (float)(NB_RINGS * q.bin);
  }
  if (u < 0.f || v < 0.f) return false;
  *bx = (int)floorf(u / (float)q.bin);
  *by = (int)floorf(v / (float)q.bin);
  return *bx < q.nbx && *by < q.nby;   // beyond the margin: further than any answer that is called final
}
__global__ __launch_bounds__(256) void k_nb_count(int64_t n, const float *__restrict__ ij, NbParams q, int32_t *__restrict__ cnt) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= n) return;
  int bx, by;
  if (nb_bin_of(q, ij[2 * c], ij[2 * c + 1], &bx, &by)) atomicAdd(&cnt[(int64_t)by * q.nbx + bx], 1);
}
__global__ __launch_bounds__(256) void k_nb_fill(int64_t n, int64_t first, const float *__restrict__ ij, NbParams q, const int32_t *__restrict__ off,
                                                 int32_t *__restrict__ cur, int32_t *__restrict__ ids) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= n) return;
  int bx, by;
  if (!nb_bin_of(q, ij[2 * c], ij[2 * c + 1], &bx, &by)) return;
  const int64_t b = (int64_t)by * q.nbx + bx;
  ids[off[b] + atomicAdd(&cur[b], 1)] = (int32_t)(first + c);
}
// one thread per target point.  out[p] = the nearest cell when that is final, -1 when the point is left to the BVH search
__global__ __launch_bounds__(256) void k_nb_query(int npx, int npy, const double *__restrict__ px, const double *__restrict__ py,
                                                  const double *__restrict__ pz, const double *__restrict__ cx, const double *__restrict__ cy,
                                                  const double *__restrict__ cz, NbParams q, const int32_t *__restrict__ off,
                                                  const int32_t *__restrict__ ids, int32_t *__restrict__ out, int32_t *__restrict__ flags) {
  const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const bool live = p < (int64_t)npx * npy;
  bool done = false;
  if (live) {
  const int i = (int)(p % npx), j = (int)(p / npx);
  const double X = px[p], Y = py[p], Z = pz[p];
  const int bx = q.per ? (i % q.nxp) / q.bin : (i + NB_RINGS * q.bin) / q.bin, by = (j + NB_RINGS * q.bin) / q.bin;
  const double r2d = 57.29577951308232, alat = fabs(asin(fmin(fmax(Z, -1.0), 1.0)));   // the point's |latitude|, radians
  // cells WITHOUT a usable index are not binned: nothing as far as they can be is final.  They sit beyond zlim degrees (lat-lon)
  // resp. within a degree of the Lambert pole or more than 60 degrees into the other hemisphere.
  double cap = q.cap;
  if (q.zone == 1 || q.zone == 4) cap = fmin(cap, 0.9 * 2.0 * sin(0.5 * fmax(q.zlim / r2d - alat, 0.0)));
  if (q.zone == 2 || q.zone == 3) {
    const double hl = q.hemi * asin(fmin(fmax(Z, -1.0), 1.0)) * r2d;   // degrees towards the projection's pole
    // (polar stereographic, zone 3: its own pole is a regular point of the map; only the far limit counts)
    cap = fmin(cap, 0.9 * 2.0 * sin(0.5 * fmax(q.zone == 2 ? fmin(89.0 - hl, hl + 60.0) : hl + 60.0, 0.0) / r2d));
  }
  if (q.zone == 2 || q.zone == 4) {
    // ... and a cell across the projection's cut (the meridian opposite the standard longitude) sits far away in index space
    // however near it is on the sphere: nothing as far as the cut is final
    double dl = atan2(Y, X) * r2d - q.stdlon;
    dl -= 360.0 * floor((dl + 180.0) / 360.0);
    if (fabs(dl) > 90.0) {
      const double ang = asin(fmin(cos(alat) * sin((180.0 - fabs(dl)) / r2d), 1.0));   // angular distance to the cut meridian
      cap = fmin(cap, 0.9 * 2.0 * sin(0.5 * ang));
    }
  }
  double best = INFINITY;
  int32_t best_id = 0x7fffffff;
  for (int r = 0; r <= NB_RINGS && !done; ++r) {
    // the bins of ring r: the square of half-width r without the square of half-width r - 1
    for (int yy = by - r; yy <= by + r; ++yy) {
      if (yy < 0 || yy >= q.nby) continue;
      const bool edge_row = yy == by - r || yy == by + r;
      for (int xx = bx - r; xx <= bx + r; xx += (edge_row || r == 0) ? 1 : 2 * r) {
        int xw = xx;
        if (q.per) xw = (xx % q.nbx + q.nbx) % q.nbx;
        else if (xx < 0 || xx >= q.nbx) continue;
        const int64_t b = (int64_t)yy * q.nbx + xw;
        for (int32_t k = off[b]; k < off[b + 1]; ++k) {
          const int32_t id = ids[k];
          const double d = dist2_nofma(X, Y, Z, cx[id], cy[id], cz[id]);
          if (d < best || (d == best && id < best_id)) {
            best = d;
            best_id = id;
          }
        }
      }
    }
    // every cell not seen so far is at least (r * bin) index units from the point (it sits in its own bin: the distance to the
    // edge of the block of rings 0 .. r is at least r bins), less the rounding of the float32 indices; on the sphere that is at
    // least that many times the chord length of an index unit -- on a lat-lon grid taken where the rings reach furthest poleward
    double h = q.h;
    if (q.local) h = 0.8 * fmin(q.dlat, q.dlon * cos(fmin(alat + (double)(r * q.bin + 1) * q.dlat, 1.5707)));
    const double lim = fmin(((double)(r * q.bin) - 2e-3) * h, cap);
    done = r > 0 && best <= lim * lim;
  }
  out[p] = done ? best_id : -1;
  }
  // points left to the tree: bit 1 of flags[0], their number in flags[1] (one atomic per workgroup)
  const int nleft = __syncthreads_count(live && !done);
  if (threadIdx.x == 0 && nleft) {
    atomicOr(flags, 2);
    atomicAdd(flags + 1, nleft);
  }
}
// cells with a usable index on the grid or within `margin` index units of it
__global__ __launch_bounds__(256) void k_nb_inside(int64_t n, const float *__restrict__ ij, float margin, float npx, float npy,
                                                   unsigned long long *__restrict__ out) {
  __shared__ unsigned long long sn;
  if (threadIdx.x == 0) sn = 0;
  __syncthreads();
  unsigned long long k = 0;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < n; c += (int64_t)gridDim.x * blockDim.x) {
    const float i = ij[2 * c], j = ij[2 * c + 1];
    k += (i >= -margin && i <= npx + margin && j >= -margin && j <= npy + margin) ? 1ull : 0ull;   // (false for NaN)
  }
  if (k) atomicAdd(&sn, k);
  __syncthreads();
  if (threadIdx.x == 0 && sn) atomicAdd(out, sn);
}
__global__ __launch_bounds__(256) void k_zrange(int64_t n, const double *__restrict__ z, unsigned long long *__restrict__ out) {
  __shared__ double slo[4], shi[4];
  double lo = 2.0, hi = -2.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    lo = fmin(lo, z[i]);
    hi = fmax(hi, z[i]);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = fmin(lo, __shfl_down(lo, o));
    hi = fmax(hi, __shfl_down(hi, o));
  }
  if ((threadIdx.x & 63) == 0) {
    slo[threadIdx.x >> 6] = lo;
    shi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // z + 2 is positive: its bit pattern orders like the value
    atomicMin(out, (unsigned long long)__double_as_longlong(fmin(fmin(slo[0], slo[1]), fmin(slo[2], slo[3])) + 2.0));
    atomicMax(out + 1, (unsigned long long)__double_as_longlong(fmax(fmax(shi[0], shi[1]), fmax(shi[2], shi[3])) + 2.0));
  }
}

// -> *state: 0 nothing usable came out (take the BVH search for all points), 1 every point settled, 2 the points whose h->idx
// entry is -1 are left to the BVH search (masked)
static int nearest_by_bins(mpg_mesh_s *m, mpg_grid_s *g, int stagger, int npx, int npy, const PointSet &pts, mpg_handle_s *h, hipStream_t s,
                           int *state) {
  *state = 0;
  int rc;
  const int64_t P = (int64_t)npx * npy, n = m->cwn, first = m->cw0;
  if (!mpg_grid_has_inverse(g, stagger) || !mpg_store_boxes() || n == 0 || stagger == MPG_STAGGERLOC_CORNER) return MPG_SUCCESS;
  const ProjDev &pr = g->proj;
  NbParams q;
  memset(&q, 0, sizeof(q));
  q.per = (g->periodic & MPG_GRID_PERIODIC_I) != 0;
  if (q.per) {   // a global lat-lon grid: the index wraps; needs the full circle in whole bins and no duplicated column
    q.nxp = pr.nxmax - pr.nxmin + 1;
    if (pr.code != MPG_PROJ_LATLON || npx != q.nxp || g->nx != q.nxp || q.nxp % NB_BIN_MIN) return MPG_SUCCESS;
  }
  q.di = stagger == MPG_STAGGERLOC_EDGE1 ? 0.5f : 0.f;
  q.dj = stagger == MPG_STAGGERLOC_EDGE2 ? 0.5f : 0.f;
  q.cap = m->cwn < m->nCells ? m->geo_margin : 4.0;
  const double latlon_limit = 89.0;   // the nearest search only PLACES points: the lat-lon inverse is good up to the last degree
  if (pr.code == MPG_PROJ_LATLON) {
    q.local = 1;
    q.dlat = fabs(pr.latinc) * 3.141592653589793 / 180.0;
    q.dlon = fabs(pr.loninc) * 3.141592653589793 / 180.0;
    q.zone = 1;
    q.zlim = latlon_limit;
  } else if (pr.code == MPG_PROJ_PS) {
    q.zone = 3;
    q.hemi = pr.hemi;
  } else if (pr.code == MPG_PROJ_MERC) {
    q.zone = 4;
    q.zlim = 85.0;
    q.stdlon = pr.lon1 + (1.0 + 0.5 * (double)g->nx - pr.knowni) * pr.dlon * (180.0 / 3.141592653589793);   // the grid's middle column
  } else {
    q.zone = 2;
    q.hemi = pr.hemi;
    q.stdlon = pr.stdlon;
  }
  TmpBuf<float> ij;
  TmpBuf<int32_t> cnt, off, ids, flags;
  TmpBuf<unsigned long long> zr;
  // (the bin counters are sized and cleared for the smallest bins before the read-back below, so that nothing but the count waits for it)
  const int64_t nbins_max = (int64_t)(q.per ? q.nxp / NB_BIN_MIN : (npx + 2 * NB_RINGS * NB_BIN_MIN + NB_BIN_MIN - 1) / NB_BIN_MIN) *
                            ((npy + 2 * NB_RINGS * NB_BIN_MIN + NB_BIN_MIN - 1) / NB_BIN_MIN);
  if (nbins_max + 1 >= 0x7fffffff) return MPG_SUCCESS;
  if ((rc = ij.alloc(2 * (size_t)n, s)) || (rc = ids.alloc((size_t)n + 1, s)) || (rc = flags.alloc(2, s)) || (rc = zr.alloc(3, s)) ||
      (rc = cnt.alloc((size_t)nbins_max + 1, s)) || (rc = off.alloc((size_t)nbins_max + 1, s)))
    return rc;
  MPG_HIP(hipMemsetAsync(flags.p, 0, 2 * sizeof(int32_t), s));
  MPG_HIP(hipMemsetAsync(cnt.p, 0, sizeof(int32_t) * (nbins_max + 1), s));
  if ((rc = mpg_k_points_ij(g, n, m->cell.x.p + first, m->cell.y.p + first, m->cell.z.p + first, ij.p, s, latlon_limit, !q.per))) return rc;
  const unsigned nbc = (unsigned)((n + 255) / 256);
  // One small read-back: how many cells sit on or around the grid (-> the bin size: about one cell per bin) and, for Lambert, the
  // latitudes the grid's points span (-> one bound for the grid)
  MPG_HIP(hipMemsetAsync(zr.p, 0xff, sizeof(unsigned long long), s));
  MPG_HIP(hipMemsetAsync(zr.p + 1, 0, 2 * sizeof(unsigned long long), s));
  if (!q.local) k_zrange<<<(unsigned)std::min<int64_t>((P + 255) / 256, 1024), 256, 0, s>>>(P, pts.z.p, zr.p);
  k_nb_inside<<<(unsigned)std::min<int64_t>(nbc, 2048), 256, 0, s>>>(n, ij.p, (float)(NB_RINGS * NB_BIN_MAX), (float)npx, (float)npy, zr.p + 2);
  MPG_HIP(hipGetLastError());
  unsigned long long hz[3];
  MPG_HIP(hipMemcpyAsync(hz, zr.p, sizeof(hz), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  {
    // about one cell per bin where the cells are: the area the counted cells cover, (npx + 2 M) x (npy + 2 M), over their number
    const double M = NB_RINGS * NB_BIN_MAX, area = ((double)npx + 2 * M) * ((double)npy + 2 * M);
    int want = (int)floor(sqrt(area / (double)std::max<unsigned long long>(hz[2], 1)));
    want = std::max(NB_BIN_MIN, std::min(NB_BIN_MAX, want));
    if (q.per)   // the full circle in whole bins
      while (want > NB_BIN_MIN && q.nxp % want) --want;
    q.bin = want;
  }
  // a regional lat-lon grid: cell indices are unwrapped around its middle column, which is only unambiguous while the grid
  // and its margin stay well short of the full circle
  if (pr.code == MPG_PROJ_LATLON && !q.per && (double)(npx + 2 * NB_RINGS * q.bin) * fabs(pr.loninc) > 300.0) return MPG_SUCCESS;
  if (pr.code == MPG_PROJ_MERC && (double)(npx + 2 * NB_RINGS * q.bin) * pr.dlon * (180.0 / 3.141592653589793) > 330.0) return MPG_SUCCESS;   // (the same on a Mercator map)
  q.nbx = q.per ? q.nxp / q.bin : (npx + 2 * NB_RINGS * q.bin + q.bin - 1) / q.bin;
  q.nby = (npy + 2 * NB_RINGS * q.bin + q.bin - 1) / q.bin;
  const int64_t nbins = (int64_t)q.nbx * q.nby;
  // (larger bins never need more counters than the smallest ones the buffers were sized for: the bin grid is npx / bin + 2 *
  // NB_RINGS + 1 bins wide at most, which falls with the bin side for every npx; a guard, not a branch anybody takes)
  if (nbins > nbins_max) return MPG_SUCCESS;
  if (!q.local) {   // Lambert: one bound for the grid, from the latitudes its points span
    double zlo, zhi;
    memcpy(&zlo, &hz[0], sizeof(double));
    memcpy(&zhi, &hz[1], sizeof(double));
    const double r2d = 180.0 / 3.141592653589793;
    const double lat_lo = asin(fmin(fmax(zlo - 2.0, -1.0), 1.0)) * r2d, lat_hi = asin(fmin(fmax(zhi - 2.0, -1.0), 1.0)) * r2d;
    q.h = 0.8 * mpg_grid_min_index_chord(g, lat_lo, lat_hi, (double)(NB_RINGS * q.bin + 1));
    if (!(q.h > 0.0)) return MPG_SUCCESS;
  }
  k_nb_count<<<nbc, 256, 0, s>>>(n, ij.p, q, cnt.p);
  if ((rc = mpg_scan_excl_i32(cnt.p, off.p, nbins + 1, s))) return rc;
  MPG_HIP(hipMemsetAsync(cnt.p, 0, sizeof(int32_t) * (nbins + 1), s));   // now the fill cursors
  k_nb_fill<<<nbc, 256, 0, s>>>(n, first, ij.p, q, off.p, cnt.p, ids.p);
  k_nb_query<<<(unsigned)((P + 255) / 256), 256, 0, s>>>(npx, npy, pts.x.p, pts.y.p, pts.z.p, m->cell.x.p, m->cell.y.p, m->cell.z.p, q, off.p, ids.p,
                                                        h->idx.p, flags.p);
  MPG_HIP(hipGetLastError());
  int32_t hflags[2] = {0, 0};
  MPG_HIP(hipMemcpyAsync(hflags, flags.p, sizeof(hflags), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  *state = (hflags[0] & 2) ? 2 : 1;
  // mpg_handle_store_stats: [1] bin side in grid points, [2] bins, [3] points the bins could not vouch for (left to the tree),
  // [4] cells on and around the grid the bin side was sized from
  h->store_stats[1] = q.bin; h->store_stats[2] = nbins; h->store_stats[3] = hflags[1]; h->store_stats[4] = (int64_t)hz[2];
  return MPG_SUCCESS;
}

int mpg_k_store_nearest(mpg_mesh_s *m, mpg_grid_s *g, int stagger, mpg_handle_s *h, hipStream_t s) {
  int rc;
  // a mesh cut to this grid (mpg_mesh_create_window) searches the cells of its window first
  const bool windowed = m->geo_grid != nullptr && m->cwn > 0 && m->cwn < m->nCells;
  PointSet &pts = g->pts[stagger];
  int npx = g->snx[stagger], npy = g->sny[stagger];
  int64_t P = (int64_t)npx * npy;
  if (pts.n != P) {
    mpg_set_error("RegridStore: destination stagger %d has no coordinates", stagger);
    return MPG_ERR_INVALID_ARG;
  }
  h->kind = MPG_KIND_FIXED;
  h->nnz_per_row = 1;
  h->n_src = m->nCells;
  h->n_dst = P;
  h->nx_dst = npx;
  h->ny_dst = npy;
  h->nnz = P;
  if ((rc = h->idx.alloc((size_t)P))) return rc;
  // a grid that knows its projection: through its index space, for every point that search can vouch for; the others (a grid
  // sticking far out of the mesh, the polar rows of a global lat-lon grid) are left to the BVH search, masked
  int state = 0;
  if ((rc = nearest_by_bins(m, g, stagger, npx, npy, pts, h, s, &state))) return rc;
  h->store_path = state;
  if (state == 1) return MPG_SUCCESS;
  if ((rc = mpg_k_build_bvh(m, s, !windowed))) return rc;
  if ((rc = nearest_search(m, npx, npy, pts, h, s, state == 2))) return rc;
  if (!m->bvh_whole) {
    // Exact?  Every cell that is NOT a site lies further than geo_margin from every point of the grid (that is how the window
    // was cut), so a point whose nearest site is within geo_margin has its true nearest cell -- ties included: a cell at the
    // same distance is within the margin too and therefore a site.  A point further from the mesh than that (a grid that
    // sticks far out of the mesh's footprint) sends the Store to the whole mesh: all cell centres are on the device.
    TmpBuf<unsigned long long> mx;
    if ((rc = mx.alloc(1, s))) return rc;
    MPG_HIP(hipMemsetAsync(mx.p, 0, sizeof(unsigned long long), s));
    k_nn_max_d2<<<(unsigned)std::min<int64_t>((P + 255) / 256, 2048), 256, 0, s>>>(P, pts.x.p, pts.y.p, pts.z.p, h->idx.p, m->cell.x.p, m->cell.y.p,
                                                                                  m->cell.z.p, mx.p);
    MPG_HIP(hipGetLastError());
    unsigned long long bits = 0;
    MPG_HIP(hipMemcpyAsync(&bits, mx.p, sizeof(bits), hipMemcpyDeviceToHost, s));
    MPG_HIP(hipStreamSynchronize(s));
    double d2;
    memcpy(&d2, &bits, sizeof(d2));
    if (!(d2 <= m->geo_margin * m->geo_margin)) {   // (the points the index search settled are within the margin by its own cap)
      if ((rc = mpg_k_build_bvh(m, s, true))) return rc;
      if ((rc = nearest_search(m, npx, npy, pts, h, s))) return rc;
    }
  }
  return MPG_SUCCESS;
}

// mpg_init loads this translation unit's code object ahead of its first launch (mpg_api.hip: warm_modules)
const void *mpg_anchor_k_store_nearest() { return (const void *)k_morton; }
