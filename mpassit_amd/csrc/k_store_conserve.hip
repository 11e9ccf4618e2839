// K4 "build_conserve_csr": first-order conservative RegridStore.
//
// Replaces ESMF_Field[Bundle]RegridStore(regridmethod=CONSERVE) at interp.F90:372,394 (snow, snowh;
// input_data.F90:840).  Semantics: w_ij = Area(src_i ^ dst_j) / Area(dst_j)
// (normType=DSTAREA) with great-circle polygon sides on the unit sphere; src polygon = Voronoi cell
// from verticesOnCell/vertex coordinates, dst polygon = the 4 CORNER-stagger points around centre
// (i,j) (model_grid.F90:784-794,959-984).  Uncovered destination cells stay 0.
//
// MI355X-native formulation: source polygons are rasterised onto the destination cells through an
// AABB pyramid over the CORNER points (same machinery as the bilinear rasteriser), in two steps: (1) one thread per
// source cell walks the pyramid and LISTS the destination cells that survive the box / bounding-sphere tests;
// (2) one thread per (source cell, destination cell) PAIR clips it, both polygon buffers in LDS, and bumps the
// destination cell's counter.  A scan (k_prims.hip) turns the counters into CSR row offsets, a scatter pass moves the pairs
// into place, and every (short) row is sorted by source id so the stored matrix and the summation order are
// deterministic.  Source cells with more candidates than the fixed-size list of step (1) holds -- polar cells under
// lat-lon slivers, or every cell of a coarse mesh under a fine grid -- are walked again by one WORKGROUP each, which
// counts and then lists their candidates at exact size; the pairs join the same pair-parallel clip.  Destination cell areas are computed once per Store, not per overlap.
// Clipping = Sutherland-Hodgman against the 4 great-circle half-spaces.
#include <cstring>


// No floating-point contraction in this translation unit (the geometry helpers of geom.h included): a weight is a ratio of
// areas that moves by 1e-13..1e-12 of itself when ONE product of an intersection point is fused into an FMA or not, and which
// product the compiler fuses changes with unrelated edits (measured in round 4: carrying a dot product from one polygon edge to
// the next changed 74 % of configuration 4's weights in their last digits).  With the products and sums as written -- the
// oracle is compiled the same way, -ffp-contract=off -- the stored matrix is a function of the source text alone.
#pragma clang fp contract(off)

#include "geom.h"
#include "mpg_internal.h"

#define CONS_MAXV 12   // max source polygon vertices handled (MPAS maxEdges is 6..10)
#define CONS_BUF (CONS_MAXV + 4)   // a convex polygon gains at most one vertex per clip plane
#define CONS_STACK 64
#define CONS_QUEUE 2048  // breadth-first node queue of the cooperative passes

// The pyramid walk of one source cell.  MODE 3 ("candidates", one thread per source cell): every destination cell that
// passes the box / bounding-sphere tests goes into the cell's slot list tmp_dst[c*CAND_CAP ..]; a cell with more than
// CAND_CAP candidates is put on the overflow list instead.  MODE 5 / 6 (one WORKGROUP per overflowed cell, its threads
// share the cell's subtrees of the pyramid): 5 counts the cell's candidates (-> cnt_src[c]), 6 writes them as pairs at
// pair_[cp][poff[c] ..] after the scan.  Every pair is clipped by k_conserve_clip_pairs below.
// (Round 2 first clipped the overflowed cells inside this walk, polygon buffers in scratch memory: 28 ms for configuration
// 2, where all 22 204 referenced 30-km cells overflow under the 3-km grid.)
#define CAND_CAP 24   // candidate destination cells per source cell kept by the candidate pass
#define CONS_SPILL 256   // candidates per overflowed cell the cooperative count pass keeps for the list pass
#define CONS_BIGBOX 128  // cells in a polygon's index box beyond which a wavefront, not a lane, enumerates it
#define CONS_COOP_NT 64  // threads of a cooperative pass's workgroup.  Measured 64 / 128 / 256 / 512 (round 4): C5 8.31 / 8.46 / 10.3 / 15.4 ms,
                         // C2 2.27 / 2.36 / 2.98 / 4.89 -- the passes are thousands of light polygons, not a few heavy ones: one wavefront each
template <int MODE>
__global__ __launch_bounds__(MODE == 3 ? 128 : CONS_COOP_NT) void k_conserve_raster(int64_t nCells, int maxEdges, const int32_t *__restrict__ voc,
                                                         const double *__restrict__ vx, const double *__restrict__ vy,
                                                         const double *__restrict__ vz, PyramidView pyr, int nx, int ny,
                                                         const double *__restrict__ qx, const double *__restrict__ qy,
                                                         const double *__restrict__ qz, const double *__restrict__ qarea,
                                                         const double *__restrict__ qsph, int32_t *__restrict__ cnt_src,
                                                         int32_t *__restrict__ tmp_dst, int32_t *__restrict__ ovf, int32_t *__restrict__ n_ovf,
                                                         uint8_t *__restrict__ flip, const int32_t *__restrict__ poff,
                                                         int32_t *__restrict__ pair_c, int32_t *__restrict__ pair_p, const float *__restrict__ vij,
                                                         float pad_coef, float pad_latlon, float e_max, int32_t *__restrict__ spill, int spill_cap) {
  // MODE 3: one thread per source cell.  MODE 5 / 6: one WORKGROUP per overflowed source cell (ovf[blockIdx.x]).
  constexpr bool COOP = MODE == 5 || MODE == 6;
  // MODE 7: one WAVEFRONT per polygon of the big-box queue (pair_c doubles as that queue in modes 3 and 7; its length is n_ovf[2])
  int64_t c = COOP ? (int64_t)ovf[blockIdx.x] : MODE == 7 ? (int64_t)pair_c[blockIdx.x] : blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= nCells) return;
  // a polygon whose candidates the candidate pass itself spilled (25 .. CONS_SPILL of them, all found through its index box) has
  // its exact count already: nothing to count
  if (MODE == 5 && cnt_src[c] != CAND_CAP + 1) return;
  if (MODE == 5 && n_ovf && threadIdx.x == 0) atomicAdd(n_ovf + 3, 1);   // (statistics: polygons the count pass walks for)
  if (spill && (int)blockIdx.x >= spill_cap && COOP) spill = nullptr;   // beyond the spill area: counted and listed by walking
  if (MODE == 6 && spill) {
    // the count pass (MODE 5) kept the first CONS_SPILL candidates of this cell: when that was all of them the list is copied,
    // not walked for again (configuration 5: the second walk of the polar polygons was a quarter of the whole Store)
    const int n = poff[c + 1] - poff[c];
    if (n <= CONS_SPILL) {
      if (n_ovf && threadIdx.x == 0) atomicAdd(n_ovf + 4, 1);   // (statistics: lists copied from the spill area)
      for (int k = threadIdx.x; k < n; k += blockDim.x) {
        pair_c[poff[c] + k] = (int32_t)c;
        pair_p[poff[c] + k] = spill[(int64_t)blockIdx.x * CONS_SPILL + k];
      }
      return;
    }
  }
  int found = 0;
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
  if (MODE == 3 && flip) flip[c] = area < 0.0;   // the clip kernel orients the polygon the same way (also for cells that overflow below)
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
  // Is destination cell (i, j) a candidate for this polygon?  Bounding sphere of the cell (k_cell_areas) against the polygon's
  // padded box -- 4 loads decide most cells before the 12 corner coordinates are touched --, then box against box, then a
  // non-degenerate cell.  The one test of the pyramid's leaves AND of the index-space boxes below.
  auto candidate = [&](int i, int j) -> bool {
    const int64_t pc = (int64_t)j * nx + i;
    {
      const double4 sph = *reinterpret_cast<const double4 *>(qsph + 4 * pc);   // one 32-byte record per cell: one line, not four
      double sx = sph.x, sy = sph.y, sz = sph.z, r2 = sph.w;
      double ddx = fmax(fmax(lo[0] - sx, sx - hi[0]), 0.0), ddy = fmax(fmax(lo[1] - sy, sy - hi[1]), 0.0),
             ddz = fmax(fmax(lo[2] - sz, sz - hi[2]), 0.0);
      if (ddx * ddx + ddy * ddy + ddz * ddz > r2) return false;
    }
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
    if (ql[0] - qp > hi[0] || qh[0] + qp < lo[0] || ql[1] - qp > hi[1] || qh[1] + qp < lo[1] || ql[2] - qp > hi[2] || qh[2] + qp < lo[2]) return false;
    // (Round 5 tried a SEPARATING-SIDE test here -- a pair whose polygon lies wholly outside one side of the quad never becomes an
    // entry -- to keep such pairs out of the clip: the boxes above already reject them; same-box A/B in profiles/r05_conserve_prefilter.md)
    return fabs(qarea[pc]) > 0.0;   // signed area of the destination quad, computed once per grid (k_cell_areas)
  };
  // O(1) candidates on a projection-built grid (round 4; MODE 3 only): the polygon's corners in the grid's index space (vij:
  // the inverse projection of every vertex, k_target_grid.hip) bound the destination cells it can meet -- cell (i, j) covers
  // index coordinates i - 0.5 .. i + 0.5 -- and every cell of that padded box takes the test above.  More candidates than the
  // list holds, a polygon near the projection's pole / cut or wider than e_max index units: the walk (and the overflow passes).
  if ((MODE == 3 || MODE == 7) && vij) {
    float imin = 1e30f, imax = -1e30f, jmin = 1e30f, jmax = -1e30f;
    bool ok = true;
    for (int j = 0, k = 0; j < maxEdges && k < CONS_MAXV; ++j) {
      int32_t v = voc[c * maxEdges + j];
      if (v <= 0) continue;
      ++k;
      const float vi = vij[2 * (int64_t)(v - 1)], vj = vij[2 * (int64_t)(v - 1) + 1];
      ok = ok && vi == vi && vj == vj;
      imin = fminf(imin, vi); imax = fmaxf(imax, vi);
      jmin = fminf(jmin, vj); jmax = fmaxf(jmax, vj);
    }
    const float E = fmaxf(imax - imin, jmax - jmin);
    if (!ok) {
      // no usable index (poleward of 85 degrees on a lat-lon grid, the projection's pole / cut): such a polygon meets MANY thin
      // destination cells, and one lane walking the pyramid for it kept its whole wavefront resident for milliseconds (C5: 4.5 ms
      // of this kernel at 15 waves per CU).  A workgroup each does it in the cooperative passes.
      const double *bx = pyr.box + 6 * pyr.off[pyr.nlev - 1];   // ... unless it does not even meet the grid's bounding box
      if (bx[0] > hi[0] || bx[3] < lo[0] || bx[1] > hi[1] || bx[4] < lo[1] || bx[2] > hi[2] || bx[5] < lo[2]) {
        cnt_src[c] = 0;
        return;
      }
      cnt_src[c] = CAND_CAP + 1;
      ovf[atomicAdd(n_ovf, 1)] = (int32_t)c;
      return;
    }
    if (E <= e_max) {
      const float pad = mpg_box_pad(E, pad_coef, pad_latlon, fmax(fabs(lo[2]), fabs(hi[2])));   // (the padded z range: a little poleward of the vertices)
      const int i0 = max((int)ceilf(imin - pad - 0.5f), 0), i1 = min((int)floorf(imax + pad + 0.5f), nx - 1);
      const int j0 = max((int)ceilf(jmin - pad - 0.5f), 0), j1 = min((int)floorf(jmax + pad + 0.5f), ny - 1);
      // A polygon whose box holds many cells (a cell of a mesh coarser than the grid, a cell next to the poles of a lat-lon grid)
      // is not enumerated by this one lane -- a few hundred wavefronts would then do all the work of the pass while the rest of
      // the chip idles (configuration 2: this kernel at 7 resident waves per busy cycle) -- but queued for MODE 7, where the
      // lanes of a wavefront share the box.
      const int bw = max(i1 - i0 + 1, 0), bh = max(j1 - j0 + 1, 0);   // (a box off the grid is empty: i0 > i1 or j0 > j1)
      if (MODE == 3 && pair_c && bw * bh > CONS_BIGBOX) {
        const int qs = atomicAdd(n_ovf + 2, 1);
        if (qs < spill_cap) {
          pair_c[qs] = (int32_t)c;
          cnt_src[c] = CAND_CAP + 1;   // (until MODE 7 has counted)
          return;
        }
      }
      if (MODE == 7) {
        // the box, 64 cells at a time in row-major order; the candidates keep that order (ballot + prefix count), the first
        // CAND_CAP of them in the polygon's list, all of them in its spill area once there are more -- the rules of the lane form below
        const int lane = (int)threadIdx.x, total = bw * bh;
        int slot = -1;
        bool complete = true;
        for (int base = 0; base < total && complete; base += 64) {
          const int q = base + lane;
          bool is = false;
          int32_t pc = 0;
          if (q < total) {
            const int i = i0 + q % bw, j = j0 + q / bw;
            is = candidate(i, j);
            pc = (int32_t)((int64_t)j * nx + i);
          }
          const unsigned long long mask = __ballot(is);
          const int nnew = __popcll(mask);
          if (nnew == 0) continue;
          const int pos = found + __popcll(mask & ((1ull << lane) - 1ull));
          if (found + nnew > CAND_CAP && slot < 0) {
            int sl = 0;
            if (lane == 0) {
              sl = atomicAdd(n_ovf, 1);
              ovf[sl] = (int32_t)c;
            }
            slot = __shfl(sl, 0);
            if (spill && slot < spill_cap) {
              __threadfence_block();
              for (int k = lane; k < min(found, CAND_CAP); k += 64) spill[(int64_t)slot * CONS_SPILL + k] = tmp_dst[c * CAND_CAP + k];
            } else {
              complete = false;
            }
          }
          if (is) {
            if (pos < CAND_CAP) tmp_dst[c * CAND_CAP + pos] = pc;
            if (slot >= 0 && complete && pos < CONS_SPILL) spill[(int64_t)slot * CONS_SPILL + pos] = pc;
          }
          found += nnew;
          if (slot >= 0 && found > CONS_SPILL) complete = false;
        }
        if (lane == 0) cnt_src[c] = (slot < 0 || complete) ? found : CAND_CAP + 1;
        return;
      }
      // Up to CAND_CAP candidates go to the polygon's own list; a polygon with more (a cell next to the poles of a lat-lon grid, a
      // cell of a mesh coarser than the grid) takes a slot of the overflow list and SPILLS the rest into that slot's area, still
      // from its box: with at most CONS_SPILL candidates its count is exact here and the cooperative count pass has nothing to do
      // for it (the list pass copies the area, as it does for a polygon the count pass spilled).  More than that, or no area
      // left: the cooperative passes walk for it.
      int slot = -1;
      bool complete = true;
      for (int j = j0; j <= j1; ++j)
        for (int i = i0; i <= i1; ++i) {
          if (!candidate(i, j)) continue;
          const int32_t pc = (int32_t)((int64_t)j * nx + i);
          if (found < CAND_CAP) {
            tmp_dst[c * CAND_CAP + found] = pc;
          } else {
            if (slot < 0) {
              slot = atomicAdd(n_ovf, 1);
              ovf[slot] = (int32_t)c;
              if (spill && slot < spill_cap)
                for (int k = 0; k < CAND_CAP; ++k) spill[(int64_t)slot * CONS_SPILL + k] = tmp_dst[c * CAND_CAP + k];
              else
                complete = false;
            }
            if (complete && found < CONS_SPILL) spill[(int64_t)slot * CONS_SPILL + found] = pc;
            else complete = false;
          }
          ++found;
          if (!complete) {
            j = j1;   // leave both loops
            break;
          }
        }
      cnt_src[c] = (slot < 0 || complete) ? found : CAND_CAP + 1;
      return;
    }
  }
  if (MODE == 7) return;   // (only polygons with a usable box are queued)
  int stack[CONS_STACK];
  // Seeds of the depth-first walk.  MODE 0: the root.  Cooperative modes: the workgroup first expands the pyramid
  // breadth-first in LDS (one node per thread and level) down to 8 x 8-cell nodes, then every thread walks its share.
  __shared__ int q_nodes[2][CONS_QUEUE];
  __shared__ int q_n[2], q_over, s_found;
  int seed_lev = pyr.nlev - 1, nseed = 1, cur = 0;
  if (COOP) {
    if (threadIdx.x == 0) { q_nodes[0][0] = 0; q_n[0] = 1; q_over = 0; s_found = 0; }
    __syncthreads();
    while (seed_lev > 1) {
      const int nq = q_n[cur];
      if (threadIdx.x == 0) q_n[cur ^ 1] = 0;
      __syncthreads();
      const int cnx = pyr.nx[seed_lev - 1], cny = pyr.ny[seed_lev - 1], pnx = pyr.nx[seed_lev];
      for (int k = threadIdx.x; k < nq; k += blockDim.x) {
        int node = q_nodes[cur][k], bi = node % pnx, bj = node / pnx;
        for (int ch = 0; ch < 4; ++ch) {
          int ci = 2 * bi + (ch & 1), cj = 2 * bj + (ch >> 1);
          if (ci >= cnx || cj >= cny) continue;
          const double *bx = pyr.box + 6 * (pyr.off[seed_lev - 1] + cj * cnx + ci);
          if (bx[0] > hi[0] || bx[3] < lo[0] || bx[1] > hi[1] || bx[4] < lo[1] || bx[2] > hi[2] || bx[5] < lo[2]) continue;
          int slot = atomicAdd(&q_n[cur ^ 1], 1);
          if (slot < CONS_QUEUE) q_nodes[cur ^ 1][slot] = cj * cnx + ci;
          else q_over = 1;
        }
      }
      __syncthreads();
      if (q_over) break;       // next level does not fit the queue: walk from the current one (correct, less parallel)
      cur ^= 1;
      --seed_lev;
      __syncthreads();
    }
    nseed = q_n[cur];
  }
  for (int sk = COOP ? (int)threadIdx.x : 0; sk < nseed; sk += COOP ? (int)blockDim.x : 1) {
  const int seed = COOP ? q_nodes[cur][sk] : 0;
  int sp = 0;
  // node boxes of the cell pyramid already include the destination cells' own bulge (k_pyr_leaf, halo mode).  Children are
  // box-tested before they are pushed (their bounds fetched together), as in k_tri_raster: one iteration per node that
  // meets the polygon's box instead of four per level; the cooperative modes' seeds were tested when they were queued.
  auto meets = [&](int lev_, int node_) -> bool {
    const double *bx = pyr.box + 6 * (pyr.off[lev_] + node_);
    return !(bx[0] > hi[0] || bx[3] < lo[0] || bx[1] > hi[1] || bx[4] < lo[1] || bx[2] > hi[2] || bx[5] < lo[2]);
  };
  // the node to visit next stays in a register; only siblings that also meet the polygon's box go on the (scratch) stack
  int cur = (COOP || meets(seed_lev, seed)) ? ((seed_lev << 26) | seed) : -1;
  for (;;) {
    if (cur < 0) {
      if (sp == 0) break;
      cur = stack[--sp];
    }
    const int e = cur;
    cur = -1;
    int lev = e >> 26, node = e & ((1 << 26) - 1);
    int nxl = pyr.nx[lev];
    int bi = node % nxl, bj = node / nxl;
    if (lev == 0) {
      int i0 = bi * MPG_PYR_B0, j0 = bj * MPG_PYR_B0;
      int i1 = min(i0 + MPG_PYR_B0, nx), j1 = min(j0 + MPG_PYR_B0, ny);
      for (int j = j0; j < j1; ++j)
        for (int i = i0; i < i1; ++i) {
          if (!candidate(i, j)) continue;
          const int64_t p = (int64_t)j * nx + i;
          if (MODE == 3) {   // candidate pass: the pair (c, p) is clipped later by k_conserve_clip_pairs, one thread per PAIR
            if (found == CAND_CAP) {
              cnt_src[c] = CAND_CAP + 1;
              ovf[atomicAdd(n_ovf, 1)] = (int32_t)c;
              return;
            }
            tmp_dst[c * CAND_CAP + found] = (int32_t)p;
            ++found;
          } else if (MODE == 5) {
            const int slot = atomicAdd(&s_found, 1);
            if (spill && slot < CONS_SPILL) spill[(int64_t)blockIdx.x * CONS_SPILL + slot] = (int32_t)p;
          } else {
            const int slot = atomicAdd(&s_found, 1);
            pair_c[poff[c] + slot] = (int32_t)c;
            pair_p[poff[c] + slot] = (int32_t)p;
          }
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
          if (cur < 0) cur = child;
          else if (sp < CONS_STACK) stack[sp++] = child;
        }
    }
  }
  }
  if (MODE == 3) cnt_src[c] = found;
  if (MODE == 5) {
    __syncthreads();
    if (threadIdx.x == 0) cnt_src[c] = s_found;
  }
}

// signed area of every destination cell (corner order i,j -> i+1,j -> i+1,j+1 -> i,j+1), once per grid
__global__ __launch_bounds__(256) void k_cell_areas(int nx, int ny, const double *__restrict__ qx, const double *__restrict__ qy,
                                                    const double *__restrict__ qz, double *__restrict__ qarea,
                                                    double *__restrict__ qsph /* [P][4]: centre xyz, radius^2 */) {
  int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p >= (int64_t)nx * ny) return;
  int i = (int)(p % nx), j = (int)(p / nx), nxc = nx + 1;
  int64_t k00 = (int64_t)j * nxc + i;
  dv3 q0 = dv3{qx[k00], qy[k00], qz[k00]}, q1 = dv3{qx[k00 + 1], qy[k00 + 1], qz[k00 + 1]},
      q2 = dv3{qx[k00 + nxc + 1], qy[k00 + nxc + 1], qz[k00 + nxc + 1]}, q3 = dv3{qx[k00 + nxc], qy[k00 + nxc], qz[k00 + nxc]};
  qarea[p] = sph_tri_area(q0, q1, q2) + sph_tri_area(q0, q2, q3);
  // bounding sphere: centre = normalised corner mean; the farthest point of a great-circle side from a point of the
  // sphere is one of its end points, so the largest corner distance bounds the whole cell
  dv3 cen = (q0 + q1) + (q2 + q3);
  double nn = sqrt(dot3(cen, cen));
  cen = nn > 0.0 ? cen * (1.0 / nn) : q0;
  double r2 = 0.0;
  dv3 d = q0 - cen; r2 = fmax(r2, dot3(d, d));
  d = q1 - cen; r2 = fmax(r2, dot3(d, d));
  d = q2 - cen; r2 = fmax(r2, dot3(d, d));
  d = q3 - cen; r2 = fmax(r2, dot3(d, d));
  *reinterpret_cast<double4 *>(qsph + 4 * p) = double4{cen.x, cen.y, cen.z, r2 * (1.0 + 1e-9) + 1e-18};
}

// most vertices of any cell (verticesOnCell is maxEdges wide -- 10 in MPAS's own files -- whatever the cells have: 6 or 7): the
// clip kernel sizes its LDS polygons by this, not by the array's width
__global__ __launch_bounds__(256) void k_max_valence(int64_t nCells, int maxEdges, const int32_t *__restrict__ voc, int32_t *__restrict__ out) {
  __shared__ int smax;
  if (threadIdx.x == 0) smax = 0;
  __syncthreads();
  int mx = 0;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < nCells; c += (int64_t)gridDim.x * blockDim.x) {
    int n = 0;
    for (int j = 0; j < maxEdges; ++j) n += voc[c * maxEdges + j] > 0;
    mx = max(mx, n);
  }
  atomicMax(&smax, mx);
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out, smax);
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

// ---- pair-parallel clipping ---------------------------------------------------------------------------------------------
// Measured on C4 (tools/store_timing.py, round 2): of the 31 ms of the one-thread-per-source-cell pass, 24 ms were the
// Sutherland-Hodgman clip -- its two polygon buffers are private arrays indexed at run time, i.e. scratch memory, ~200
// vector-memory operations per clipped pair -- and 7 ms everything else.  So the pass is split: the traversal only LISTS
// the candidate pairs (source cell, destination cell) that survive the box / bounding-sphere tests (k_conserve_raster<3>),
// and one thread per PAIR clips it (balanced: no lane waits for a neighbour with more candidates) with both polygon
// buffers in LDS, laid out [buffer][vertex][component][lane] so that any per-lane vertex index is conflict-free.
// The arithmetic is the oracle's (oracle/mpassit_oracle.c, clip planes with normals in difference form a x (b - a)): that
// is the stated reference of the weights, not the round-1 kernel.
__global__ __launch_bounds__(256) void k_conserve_clamp_counts(int64_t nCells, const int32_t *__restrict__ cnt_src, int32_t *__restrict__ npair) {
  int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c > nCells) return;
  int n = c < nCells ? cnt_src[c] : 0;
  npair[c] = n;                          // overflowed cells: their exact count from the cooperative count pass
}
__global__ __launch_bounds__(256) void k_conserve_fill_pairs(int64_t nCells, const int32_t *__restrict__ npair, const int32_t *__restrict__ poff,
                                                             const int32_t *__restrict__ tmp_dst, int32_t *__restrict__ pair_c,
                                                             int32_t *__restrict__ pair_p) {
  int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= nCells) return;
  const int n = npair[c], o = poff[c];
  if (n > CAND_CAP) return;              // overflowed cell: its pairs are written by k_conserve_raster<6>
  for (int e = 0; e < n; ++e) {
    pair_c[o + e] = (int32_t)c;
    pair_p[o + e] = tmp_dst[c * CAND_CAP + e];
  }
}

#define CLIP_NT 64
struct LdsPoly {   // vertex i of polygon buffer `buf` of this lane
  double *base;    // lds + lane
  int cb;
  __device__ __forceinline__ dv3 get(int buf, int i) const {
    const double *p = base + (size_t)((buf * cb + i) * 3) * CLIP_NT;
    return dv3{p[0], p[CLIP_NT], p[2 * CLIP_NT]};
  }
  __device__ __forceinline__ void set(int buf, int i, dv3 v) const {
    double *p = base + (size_t)((buf * cb + i) * 3) * CLIP_NT;
    p[0] = v.x;
    p[CLIP_NT] = v.y;
    p[2 * CLIP_NT] = v.z;
  }
};
// Sutherland-Hodgman step IN PLACE, the polygon in LDS.  The output of edge i goes to slots <= i + 1 and those have been
// read by then: vertex i+1 is in registers as X2, and a CONVEX polygon meets the plane at most twice with at least one
// vertex outside between the two crossings, so the write index never passes i + 1 (the first vertex, needed again for the
// closing edge, is kept in registers).  One buffer instead of two halves the LDS of the clip kernel: 8 instead of 4
// wavefronts per CU.  The arithmetic and its order are those of the two-buffer form: the same bits.  A non-convex cell can
// break the bound or outgrow `cap` = maxEdges + 4 slots; either is reported through *trunc (the Store then fails with
// MPG_ERR_OVERFLOW), never a silently wrong polygon.
__device__ __forceinline__ int clip_halfspace_lds(int n, const LdsPoly &L, dv3 nrm, int cap, int *trunc) {
  int m = 0;
  double eps = 1e-15 * sqrt(dot3(nrm, nrm));
  const dv3 first = L.get(0, 0);
  dv3 X1 = first;
  const double dfirst = dot3(nrm, first);
  double d1 = dfirst;
  for (int i = 0; i < n; ++i) {
    const dv3 X2 = (i + 1 == n) ? first : L.get(0, i + 1);
    const double d2 = (i + 1 == n) ? dfirst : dot3(nrm, X2);   // (the same product as the next edge's d1: computed once)
    bool in1 = d1 >= -eps, in2 = d2 >= -eps;
    if (in1) {
      if (m < cap && m <= i + 1) L.set(0, m++, X1);
      else *trunc = 1;
    }
    if (in1 != in2) {
      dv3 X = X1 * d2 - X2 * d1;
      double sgn = (d2 - d1) > 0.0 ? 1.0 : -1.0;
      double nn = sqrt(dot3(X, X));
      if (nn > 0.0) {
        if (m < cap && m <= i + 1) L.set(0, m++, X * (sgn / nn));
        else *trunc = 1;
      }
    }
    X1 = X2;
    d1 = d2;
  }
  return m;
}
__global__ __launch_bounds__(CLIP_NT) void k_conserve_clip_pairs(int64_t npairs, const int32_t *__restrict__ pair_c, const int32_t *__restrict__ pair_p,
                                                                 int maxEdges, const int32_t *__restrict__ voc, const double *__restrict__ vx,
                                                                 const double *__restrict__ vy, const double *__restrict__ vz,
                                                                 const uint8_t *__restrict__ flip, int nx, const double *__restrict__ qx,
                                                                 const double *__restrict__ qy, const double *__restrict__ qz,
                                                                 const double *__restrict__ qarea, int cb, double *__restrict__ pair_val,
                                                                 int32_t *__restrict__ count, int32_t *__restrict__ truncated) {
  extern __shared__ double clip_lds[];   // [cb][3][CLIP_NT]
  const int64_t t = blockIdx.x * (int64_t)CLIP_NT + threadIdx.x;
  if (t >= npairs) return;
  const LdsPoly L{clip_lds + threadIdx.x, cb};
  const int64_t c = pair_c[t], p = pair_p[t];
  // the source polygon, counter-clockwise seen from outside (orientation decided once per cell by the candidate pass).  All of a
  // cell's vertex numbers are fetched before any coordinate and all coordinates before the first is used: written as a loop over
  // the vertices (number, then coordinates, then the next number) the gather was a chain of two memory latencies per vertex, a
  // dozen in a row, in a kernel that holds two or three wavefronts per SIMD.  Padding entries re-load the cell's first vertex.
  int32_t vid[CONS_MAXV];
#pragma unroll
  for (int k = 0; k < CONS_MAXV; ++k) vid[k] = k < maxEdges ? voc[c * maxEdges + k] : 0;
  const bool rev = flip[c] != 0;
  int ntot = 0, vsafe = 0;
#pragma unroll
  for (int k = 0; k < CONS_MAXV; ++k) {
    if (vid[k] > 0) {
      ++ntot;
      if (vsafe == 0) vsafe = vid[k];
    }
  }
  int n = 0;
  if (vsafe > 0) {
    dv3 vc[CONS_MAXV];
#pragma unroll
    for (int k = 0; k < CONS_MAXV; ++k) {
      const int64_t v = (vid[k] > 0 ? vid[k] : vsafe) - 1;
      vc[k] = dv3{vx[v], vy[v], vz[v]};
    }
#pragma unroll
    for (int k = 0; k < CONS_MAXV; ++k)
      if (vid[k] > 0) {
        L.set(0, rev ? ntot - 1 - n : n, vc[k]);
        ++n;
      }
  }
  const int i = (int)(p % nx), j = (int)(p / nx), nxc = nx + 1;
  const int64_t k00 = (int64_t)j * nxc + i;
  dv3 q[4] = {dv3{qx[k00], qy[k00], qz[k00]}, dv3{qx[k00 + 1], qy[k00 + 1], qz[k00 + 1]},
              dv3{qx[k00 + nxc + 1], qy[k00 + nxc + 1], qz[k00 + nxc + 1]}, dv3{qx[k00 + nxc], qy[k00 + nxc], qz[k00 + nxc]}};
  double aq = qarea[p];
  if (aq < 0.0) { dv3 tq = q[1]; q[1] = q[3]; q[3] = tq; aq = -aq; }
  double ar = 0.0;
  int trunc = 0;
  if (aq > 0.0) {
    const int cur = 0;
    for (int e = 0; e < 4 && n >= 3; ++e) {
      const dv3 qa = e == 0 ? q[0] : e == 1 ? q[1] : e == 2 ? q[2] : q[3];
      const dv3 qb = e == 0 ? q[1] : e == 1 ? q[2] : e == 2 ? q[3] : q[0];
      dv3 side = qb - qa;
      if (dot3(side, side) < 1e-24) continue;     // collapsed side (pole): bounds nothing
      // a x (b - a) = a x b in difference form: the direct product's rounding would shift the plane by 1e-16 / |b - a| radians
      n = clip_halfspace_lds(n, L, cross3(qa, side), cb, &trunc);
    }
    if (n >= 3) {
      double sa = 0.0;
      const dv3 v0 = L.get(cur, 0);
      for (int k = 1; k + 1 < n; ++k) sa += sph_tri_area(v0, L.get(cur, k), L.get(cur, k + 1));
      ar = sa > 0.0 ? sa : 0.0;
    }
  }
  double ratio = 0.0;
  if (ar > 1e-14 * aq) {
    ratio = ar / aq;
    atomicAdd(&count[p], 1);
  }
  pair_val[t] = ratio;
  if (trunc) atomicOr(truncated, 1);
}
__global__ __launch_bounds__(256) void k_conserve_scatter_pairs(int64_t npairs, const int32_t *__restrict__ pair_c, const int32_t *__restrict__ pair_p,
                                                                const double *__restrict__ pair_val, const int32_t *__restrict__ rowptr,
                                                                int32_t *__restrict__ cursor, int32_t *__restrict__ col, double *__restrict__ val,
                                                                int32_t cell0) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= npairs) return;
  const double r = pair_val[t];
  if (!(r > 0.0)) return;
  const int32_t p = pair_p[t];
  const int slot = atomicAdd(&cursor[p], 1);
  col[rowptr[p] + slot] = pair_c[t] + cell0;   // the walk and the clip work on the rows of the mesh's geometry window; the matrix holds global ids
  val[rowptr[p] + slot] = r;
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
  TmpBuf<int32_t> count, cnt_src, tmp_dst, ovf, n_ovf, npair, poff, pair_c, pair_p;
  TmpBuf<double> qarea, qsph, pair_val;
  TmpBuf<uint8_t> flip;
  // source cells = the rows of the mesh's geometry window (all of them unless the mesh was cut to this grid, mpg_mesh_create_window);
  // the kernels number them 0 .. nC - 1, vertex coordinates are reached through pointers biased by the window's first vertex
  const int64_t nC = m->cwn;
  const double *vx = m->vx_g(), *vy = m->vy_g(), *vz = m->vz_g();
  if (nC == 0) {   // a window without cells (the grid lies off the mesh): the empty matrix
    if ((rc = h->rowptr.alloc((size_t)P + 1)) || (rc = h->col.alloc(1)) || (rc = h->val.alloc(1))) return rc;
    MPG_HIP(hipMemsetAsync(h->rowptr.p, 0, sizeof(int32_t) * (P + 1), s));
    MPG_HIP(hipStreamSynchronize(s));
    h->nnz = 0;
    return MPG_SUCCESS;
  }
  if ((rc = count.alloc((size_t)P + 1, s)) || (rc = h->rowptr.alloc((size_t)P + 1)) || (rc = qarea.alloc((size_t)P, s)) || (rc = qsph.alloc(4 * (size_t)P, s)) ||
      (rc = cnt_src.alloc((size_t)nC, s)) || (rc = tmp_dst.alloc((size_t)nC * CAND_CAP, s)) || (rc = ovf.alloc((size_t)nC, s)) || (rc = n_ovf.alloc(5, s)) ||
      (rc = npair.alloc((size_t)nC + 1, s)) || (rc = poff.alloc((size_t)nC + 1, s)) || (rc = flip.alloc((size_t)nC, s)))
    return rc;
  MPG_HIP(hipMemsetAsync(n_ovf.p, 0, 5 * sizeof(int32_t), s));   // [0] overflowed cells, [1] the largest vertex count of a cell, [2] big-box queue,
                                                                  // [3] polygons the count pass walked for, [4] lists the list pass copied (statistics)
  MPG_HIP(hipMemsetAsync(cnt_src.p, 0, sizeof(int32_t) * (size_t)nC, s));  // degenerate cells leave early
  MPG_HIP(hipMemsetAsync(flip.p, 0, (size_t)nC, s));
  MPG_HIP(hipMemsetAsync(count.p, 0, sizeof(int32_t) * (P + 1), s));
  unsigned nb = (unsigned)((nC + 127) / 128);
  PyramidView pv = mpg_pyr_view(g->cellpyr);
  // a grid built from its projection: the vertices' places in its index space (one inverse projection per vertex, shared by the
  // three cells around it) give every polygon its candidate cells in O(1); "store_boxes" 0 keeps the pyramid walk (A/B)
  TmpBuf<float> vij;
  const float *vijp = nullptr;
  if (mpg_grid_has_inverse(g, MPG_STAGGERLOC_CORNER) && mpg_store_boxes() && m->vwn > 0) {
    if ((rc = vij.alloc(2 * (size_t)m->vwn, s))) return rc;
    if ((rc = mpg_k_points_ij(g, m->vwn, m->vert.x.p, m->vert.y.p, m->vert.z.p, vij.p, s))) return rc;
    vijp = vij.p - 2 * m->vw0;   // indexed with global vertex ids, like vx / vy / vz
    h->store_path = 1;
  }
  k_cell_areas<<<(unsigned)((P + 255) / 256), 256, 0, s>>>(nx, ny, cor.x.p, cor.y.p, cor.z.p, qarea.p, qsph.p);
  // (1) candidate pairs: one thread per source cell walks the pyramid and lists the destination cells that pass the tests
  // the spill areas of the overflow list's first slots (CONS_SPILL candidates each): filled by the candidate pass for polygons it
  // can enumerate from their index boxes, by the cooperative count pass for the others
  const int spill_cap = (int)std::min<int64_t>(nC, 1 << 16);
  TmpBuf<int32_t> spill, bigq;
  if ((rc = spill.alloc((size_t)spill_cap * CONS_SPILL, s)) || (rc = bigq.alloc((size_t)spill_cap, s))) return rc;
  k_conserve_raster<3><<<nb, 128, 0, s>>>(nC, m->maxEdges, m->voc.p, vx, vy, vz, pv, nx, ny, cor.x.p, cor.y.p, cor.z.p,
                                        qarea.p, qsph.p, cnt_src.p, tmp_dst.p, ovf.p, n_ovf.p, flip.p, nullptr, vijp ? bigq.p : nullptr, nullptr, vijp,
                                        (float)mpg_grid_box_pad_coef(g), (float)mpg_grid_box_pad_latlon(g), (float)mpg_grid_box_emax(g), spill.p, spill_cap);
  if (m->max_valence < 0) k_max_valence<<<(unsigned)std::min<int64_t>((nC + 255) / 256, 2048), 256, 0, s>>>(nC, m->maxEdges, m->voc.p, n_ovf.p + 1);
  MPG_HIP(hipGetLastError());
  int32_t novf = 0, hv[3] = {0, 0, 0};
  MPG_HIP(hipMemcpyAsync(hv, n_ovf.p, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  novf = hv[0];
  if (m->max_valence < 0) m->max_valence = hv[1];
  if (hv[2] > 0) {   // polygons with big index boxes: a wavefront each (they may add to the overflow list: read its length again)
    k_conserve_raster<7><<<(unsigned)std::min(hv[2], spill_cap), 64, 0, s>>>(nC, m->maxEdges, m->voc.p, vx, vy, vz, pv, nx, ny, cor.x.p, cor.y.p, cor.z.p,
                                                                            qarea.p, qsph.p, cnt_src.p, tmp_dst.p, ovf.p, n_ovf.p, nullptr, nullptr, bigq.p, nullptr,
                                                                            vijp, (float)mpg_grid_box_pad_coef(g), (float)mpg_grid_box_pad_latlon(g),
                                                                            (float)mpg_grid_box_emax(g), spill.p, spill_cap);
    MPG_HIP(hipGetLastError());
    MPG_HIP(hipMemcpyAsync(&novf, n_ovf.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MPG_HIP(hipStreamSynchronize(s));
  }
  const unsigned coop_nt = CONS_COOP_NT;
  if (novf > 0)   // cells with more candidates than their list holds: one workgroup each counts them exactly
    k_conserve_raster<5><<<(unsigned)novf, coop_nt, 0, s>>>(nC, m->maxEdges, m->voc.p, vx, vy, vz, pv, nx, ny, cor.x.p,
                                                       cor.y.p, cor.z.p, qarea.p, qsph.p, cnt_src.p, nullptr, ovf.p, n_ovf.p, nullptr, nullptr,
                                                       nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, spill.p, spill_cap);
  // (2) pair list: offsets by scan, then (cell, destination) per pair
  k_conserve_clamp_counts<<<(unsigned)((nC + 256) / 256), 256, 0, s>>>(nC, cnt_src.p, npair.p);
  if ((rc = mpg_scan_excl_i32(npair.p, poff.p, nC + 1, s))) return rc;
  // the pair count twice -- the int32 scan's last entry and a 64-bit sum (the scan could wrap more than once) -- in ONE round trip
  int32_t npairs = 0;
  long long total = 0;
  {
    TmpBuf<long long> tot;
    if ((rc = tot.alloc(1, s))) return rc;
    if ((rc = mpg_sum_i32_i64(npair.p, nC, tot.p, s))) return rc;
    MPG_HIP(hipMemcpyAsync(&npairs, poff.p + nC, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MPG_HIP(hipMemcpyAsync(&total, tot.p, sizeof(total), hipMemcpyDeviceToHost, s));
    MPG_HIP(hipStreamSynchronize(s));
    if (npairs < 0 || total != (long long)npairs) {
      mpg_set_error("conservative RegridStore: %lld candidate pairs exceed 2^31", total);
      return MPG_ERR_OVERFLOW;
    }
  }
  if ((rc = pair_c.alloc((size_t)npairs + 1, s)) || (rc = pair_p.alloc((size_t)npairs + 1, s)) || (rc = pair_val.alloc((size_t)npairs + 1, s))) return rc;
  k_conserve_fill_pairs<<<(unsigned)((nC + 255) / 256), 256, 0, s>>>(nC, npair.p, poff.p, tmp_dst.p, pair_c.p, pair_p.p);
  if (novf > 0)
    k_conserve_raster<6><<<(unsigned)novf, coop_nt, 0, s>>>(nC, m->maxEdges, m->voc.p, vx, vy, vz, pv, nx, ny, cor.x.p,
                                                       cor.y.p, cor.z.p, qarea.p, qsph.p, nullptr, nullptr, ovf.p, n_ovf.p, nullptr, poff.p,
                                                       pair_c.p, pair_p.p, nullptr, 0.f, 0.f, 0.f, spill.p, spill_cap);
  MPG_HIP(hipGetLastError());
  // (3) clip: one thread per pair, polygon buffers in LDS; counts the entries per destination cell
  // buffer slots per polygon: the in-place step never lets a polygon gain more than one vertex per half-space (it reports the
  // polygon otherwise), so the cells' largest vertex count + 4 is all that can be used; 24 bytes x 64 lanes each, and the slots
  // decide how many wavefronts a CU holds (10 slots: ten, 16: six -- configuration 4's clip 1.55 -> 1.3 ms from 12 to 10)
  const int nv = std::max(3, std::min(m->max_valence, m->maxEdges));
  const int cb = nv + 4 < CONS_BUF ? nv + 4 : CONS_BUF;
  const size_t clip_lds_bytes = sizeof(double) * cb * 3 * CLIP_NT;
  if (clip_lds_bytes > 48 * 1024)
    MPG_HIP(hipFuncSetAttribute((const void *)k_conserve_clip_pairs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)clip_lds_bytes));
  TmpBuf<int32_t> truncated;
  if ((rc = truncated.alloc(1, s))) return rc;
  MPG_HIP(hipMemsetAsync(truncated.p, 0, sizeof(int32_t), s));
  if (npairs > 0)
    k_conserve_clip_pairs<<<(unsigned)(((int64_t)npairs + CLIP_NT - 1) / CLIP_NT), CLIP_NT, clip_lds_bytes, s>>>(
        npairs, pair_c.p, pair_p.p, m->maxEdges, m->voc.p, vx, vy, vz, flip.p, nx, cor.x.p, cor.y.p, cor.z.p, qarea.p, cb,
        pair_val.p, count.p, truncated.p);
  MPG_HIP(hipGetLastError());
  if ((rc = mpg_scan_excl_i32(count.p, h->rowptr.p, P + 1, s))) return rc;
  int32_t nnz = 0, was_truncated = 0, hs[5] = {0, 0, 0, 0, 0};
  MPG_HIP(hipMemcpyAsync(hs, n_ovf.p, sizeof(hs), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipMemcpyAsync(&nnz, h->rowptr.p + P, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipMemcpyAsync(&was_truncated, truncated.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MPG_HIP(hipStreamSynchronize(s));
  if (was_truncated) {
    mpg_set_error("conservative RegridStore: a clipped polygon outgrew its %d vertex slots (a non-convex source cell?)", cb);
    return MPG_ERR_OVERFLOW;
  }
  if (nnz < 0) {
    mpg_set_error("conservative weight matrix exceeds 2^31 entries");
    return MPG_ERR_OVERFLOW;
  }
  h->nnz = nnz;
  // mpg_handle_store_stats: [1] pairs clipped, [2] polygons that outgrew their list, [3] polygons enumerated by a wavefront,
  // [4] polygons the cooperative count pass walked the pyramid for, [5] lists copied from the spill area, [6] polygon slots of the clip
  h->store_stats[1] = npairs; h->store_stats[2] = novf; h->store_stats[3] = std::min(hv[2], spill_cap); h->store_stats[4] = hs[3];
  h->store_stats[5] = hs[4]; h->store_stats[6] = cb;
  if ((rc = h->col.alloc((size_t)nnz + 1)) || (rc = h->val.alloc((size_t)nnz + 1))) return rc;
  MPG_HIP(hipMemsetAsync(count.p, 0, sizeof(int32_t) * (P + 1), s));
  if (npairs > 0)
    k_conserve_scatter_pairs<<<(unsigned)(((int64_t)npairs + 255) / 256), 256, 0, s>>>(npairs, pair_c.p, pair_p.p, pair_val.p, h->rowptr.p, count.p,
                                                                                      h->col.p, h->val.p, (int32_t)m->cw0);
  k_csr_sort_rows<<<(unsigned)((P + 255) / 256), 256, 0, s>>>(P, h->rowptr.p, h->col.p, h->val.p);
  MPG_HIP(hipGetLastError());
  MPG_HIP(hipStreamSynchronize(s));
  return MPG_SUCCESS;
}

// mpg_init loads this translation unit's code object ahead of its first launch (mpg_api.hip: warm_modules)
const void *mpg_anchor_k_store_conserve() { return (const void *)k_cell_areas; }
