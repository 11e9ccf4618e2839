// The wind chain of interp_hist_data as ONE pass over the mass-point winds (SURVEY s2.2 K6 + K7):
//
//   rotate_winds_cgrid(u_target_grid_nostag, v_target_grid_nostag)            interp.F90:291-293, 689-749
//   ESMF_FieldRegridStore / Regrid  UMASS(CENTER) -> U(EDGE1)                 interp.F90:295-311
//   ESMF_FieldRegridStore / Regrid  VMASS(CENTER) -> V(EDGE2)                 interp.F90:313-328
//
// The reference (and this library until round 5: k_rotate + 2 x k_applyN<4>) makes three passes: the rotation reads and
// rewrites both mass fields, each destaggering reads one of them again.  u/v_target_grid_nostag are intermediates the
// reference never writes to its file (write_data.F90 has no `nostag`), so here the earth-relative mass winds are read ONCE:
// a workgroup owns 64 x 16 points of the (i, j) index space, stages the (64 + 2 + WD_A - 1) x (16 + 2) window of mass points
// around them in LDS -- rotated on the way in, with the operation sequence of interp.F90:737-748 (no contraction, as
// k_rotate: bit-identical) -- and combines every U point (EDGE1, nx + 1 columns) and every V point (EDGE2, ny + 1 rows) of
// its tile from there with the 4-point weights of the two Grid -> Grid handles, in k_applyN's accumulation order (the same
// bits as the three-pass chain).  Algorithmic bytes per level: 2 x 8 B read + 2 x e_dst written per mass point (+ 2 x 48 B
// of indices and weights and 16 B of rotation angles per point, once per launch) against 4 x 8 + 2 x (8 + e_dst) before.
//
// Store segments: a wave stores 64 consecutive points of ONE output row, and each row's segment is shifted left so that it
// starts on a multiple of WD_A = 8 elements in memory whatever the row length (U rows are nx + 1 = 1801 wide on the README's
// grid: unshifted, every segment would straddle one more line and leave two partial lines to another workgroup); the 7
// extra window columns pay for that.  The window is double-buffered (one barrier per level) and level k + 1 is in flight
// in registers while level k is combined.
// Generality: the handles are ordinary Grid -> Grid handles (k_store_gridbil.hip); a point whose four sources do not all
// lie in its tile's window (none on the grids of the Store above; kept for handles of other origin) is combined from global
// memory after the level loop.  Periodic grids (global lat-lon, no rotation there: interp.F90:291 asks for PROJ_LC) wrap
// the window's columns; their pole caps are rewritten afterwards by k_pole_fix exactly as after k_applyN.
#pragma clang fp contract(off)
#include "geom.h"
#include "mpg_internal.h"

#define WD_TX 64
#ifndef WD_TY
#define WD_TY 16
#endif
#define WD_NT 512
#ifndef WD_A
#define WD_A 8                          // store segments start on multiples of 8 elements (A/B r06: 1, 8, 16, 32 within 4 %, 8 ahead)
#endif
#define WD_WW (WD_TX + 2 + WD_A - 1)    // 73 window columns: i0 - WD_A .. i0 + 64
#define WD_WH (WD_TY + 2)               // 18 window rows:    j0 - 1  .. j0 + 16
#define WD_NW (WD_WW * WD_WH)
#define WD_WPT ((WD_NW + WD_NT - 1) / WD_NT)
#define WD_RPT (WD_TX * WD_TY / WD_NT)  // output rows per thread and field

struct WindArgs {
  const int32_t *idx1, *idx2;   // [4][(nx+1) * ny], [4][nx * (ny+1)]: linear CENTER indices, -1 = unmapped
  const double *w1, *w2;
  const double *cosa, *sina;    // [ny][nx] (ROT only)
  const double *um, *vm;        // [nlev][ny][nx]
  void *u, *v;                  // [nlev][ny][nx+1], [nlev][ny+1][nx]
  double *um_rot, *vm_rot;      // optional: the rotated mass winds [nlev][ny][nx] (never the inputs themselves)
  int nx, ny, nlev, periodic, mode, ntx;
  double scale, offset;
  int dbe;
};

// interp.F90:737-748 as written (k_rotate of k_apply.hip evaluates the same expressions in the same order)
__device__ __forceinline__ void wd_rotate(double uo, double vo, double ca, double sa, double tana, double den, double &un, double &vn) {
  const double t1 = vo * tana;
  un = (uo + t1) / den;
  const double t2 = un * sa;
  vn = (vo - t2) / ca;
}

// KEEP: the rotated mass winds are written too (um_rot / vm_rot) -- a template parameter, because its per-lane condition is a divergent
// branch around a store, and with one in the level loop the compiler waits for EVERY outstanding store before it consumes the next
// level's loads (vmcnt(0)); without, it waits for the loads only (the four stores of the level stay in flight)
template <typename TD, bool ROT, bool TYPED, bool KEEP>
__global__ __launch_bounds__(WD_NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_wind_destagger(const WindArgs a) {
  __shared__ double lds[2 * 2 * WD_NW];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = (int)(lin % (unsigned)a.ntx), ty = (int)(lin / (unsigned)a.ntx);
  const int nx = a.nx, ny = a.ny, nlev = a.nlev;
  const int i0 = tx * WD_TX, j0 = ty * WD_TY, wi0 = i0 - WD_A, wj0 = j0 - 1;
  const int64_t NP = (int64_t)nx * ny;
  const bool per = a.periodic != 0, do_u = (a.mode & 1) != 0, do_v = (a.mode & 2) != 0;
  const bool src_u = ROT || do_u, src_v = ROT || do_v;
  const Swz zd = make_swz(a.dbe);
  // U / V segments start on multiples of WD_A = 8 elements of a row -- 64 bytes (float64) / 32 bytes (float32) into a 128-byte line at
  // best, and U / V planes are not whole numbers of lines (1801 x 1060, 1800 x 1061): every level's runs have partial lines at their
  // ends.  Stored per LANE (geom.h buf_store_lane: whole lines non-temporal, the ends write-back so that the neighbour's part meets
  // them in L2): no branch, both stores under complementary lane masks -- a branch around stores in this loop would make every level
  // wait for the stores before it (below).
  const unsigned lane_bytes = (unsigned)lane * (unsigned)sizeof(TD);

  // ---- this thread's window points: byte offset inside a level plane (MPG_BUF_NONE: outside the grid -> loads 0) ----
  uint32_t gb[WD_WPT];
  bool wown[WD_WPT];
  double rca[WD_WPT], rsa[WD_WPT], rtn[WD_WPT], rdn[WD_WPT];
#pragma unroll
  for (int r = 0; r < WD_WPT; ++r) {
    const int e = t + WD_NT * r, ly = e / WD_WW, lx = e - ly * WD_WW;
    const int cj = wj0 + ly;
    int ci = wi0 + lx;
    const bool raw_in = ci >= 0 && ci < nx;
    if (per) ci = ((ci % nx) + nx) % nx;
    const bool val = e < WD_NW && cj >= 0 && cj < ny && ci >= 0 && ci < nx;
    wown[r] = val && raw_in && lx >= WD_A && lx < WD_A + WD_TX && ly >= 1 && ly <= WD_TY;
    const int32_t g = val ? cj * nx + ci : 0;
    gb[r] = val ? (uint32_t)g * 8u : MPG_BUF_NONE;
    rca[r] = 1.0; rsa[r] = 0.0; rtn[r] = 0.0; rdn[r] = 1.0;
    if constexpr (ROT) {
      if (val) {
        rca[r] = a.cosa[g];
        rsa[r] = a.sina[g];
        rtn[r] = rsa[r] / rca[r];
        rdn[r] = rca[r] + rsa[r] * rtn[r];
      }
    }
  }

  // ---- this thread's output points: WD_RPT rows of U and of V.  A point's four sources are the corners of one quad of
  // CENTER points (k_store_gridbil.hip: A, B = A + 1, C = B + nx, D = A + nx, columns wrapping on a periodic grid): the slot of A
  // in the window stands for all four; anything else (pole-cap fillers, handles of other origin) is a `far` point ----
  int lu[WD_RPT], lv[WD_RPT];
  double wu[WD_RPT][4], wv[WD_RPT][4];
  bool mapu[WD_RPT], mapv[WD_RPT], faru[WD_RPT], farv[WD_RPT];
  uint32_t pu[WD_RPT], pv[WD_RPT];     // byte offset inside a level plane of U / V; MPG_BUF_NONE: nothing to store in the level loop
  const int nxu = nx + 1;
  const int64_t P1 = (int64_t)nxu * ny, P2 = (int64_t)nx * (ny + 1);
  auto locate = [&](int32_t c, bool &far) -> int {   // global CENTER index -> slot of the window
    const int cj = c / nx, ci = c - cj * nx;
    int lx = ci - wi0;
    const int ly = cj - wj0;
    if (per) {
      if (lx < 0) lx += nx;
      else if (lx >= WD_WW) lx -= nx;
    }
    const bool in = lx >= 0 && lx < WD_WW && ly >= 0 && ly < WD_WH;
    far = far || !in;
    return in ? ly * WD_WW + lx : 0;
  };
  auto point = [&](const int32_t *idx, const double *w, int64_t P, int64_t p, int &slot, double *w4, bool &mapped, bool &far) {
    int32_t c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      c[q] = idx[q * P + p];
      w4[q] = w[q * P + p];
    }
    mapped = c[0] >= 0;
    far = false;
    slot = 0;
    if (mapped) {
      slot = locate(c[0], far);
      const int l1 = locate(c[1], far), l2 = locate(c[2], far), l3 = locate(c[3], far);
      far = far || l1 != slot + 1 || l2 != slot + WD_WW + 1 || l3 != slot + WD_WW;
      if (far) slot = 0;
    }
  };
#pragma unroll
  for (int r = 0; r < WD_RPT; ++r) {
    const int j = j0 + wave + (WD_NT / 64) * r;
    {
      const int i = i0 + lane - (int)(((long long)j * nxu) % WD_A);
      const bool act = do_u && j < ny && i >= 0 && i < nxu;
      const int64_t p = act ? (int64_t)j * nxu + i : 0;
      lu[r] = 0; mapu[r] = false; faru[r] = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) wu[r][q] = 0.0;
      if (act) point(a.idx1, a.w1, P1, p, lu[r], wu[r], mapu[r], faru[r]);
      pu[r] = (act && !faru[r]) ? (uint32_t)p * (uint32_t)sizeof(TD) : MPG_BUF_NONE;
    }
    {
      const int i = i0 + lane - (int)(((long long)j * nx) % WD_A);
      const bool act = do_v && j <= ny && i >= 0 && i < nx;
      const int64_t p = act ? (int64_t)j * nx + i : 0;
      lv[r] = 0; mapv[r] = false; farv[r] = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) wv[r][q] = 0.0;
      if (act) point(a.idx2, a.w2, P2, p, lv[r], wv[r], mapv[r], farv[r]);
      pv[r] = (act && !farv[r]) ? (uint32_t)p * (uint32_t)sizeof(TD) : MPG_BUF_NONE;
    }
  }

  auto finish = [&](double val, bool mapped) -> TD {
    val = mapped ? val : 0.0;
    if constexpr (TYPED) return swz<true>((TD)fma(val, a.scale, a.offset), zd);
    else return (TD)val;
  };

  // ---- level loop: branch-free (buffer addressing: lanes without a point load 0 / have their store dropped), so that the
  // wait for level k + 1's loads does not also wait for level k's stores (one in-order counter on gfx950) -------------
  const uint32_t src_bytes = (uint32_t)(NP * 8), u_bytes = (uint32_t)(P1 * sizeof(TD)), v_bytes = (uint32_t)(P2 * sizeof(TD));
  double fu[WD_WPT], fv[WD_WPT];
  auto fetch = [&](int k) {
    const BufRsrc ru = buf_rsrc(a.um + (src_u ? (int64_t)k * NP : 0), src_u ? src_bytes : 0u);
    const BufRsrc rv = buf_rsrc(a.vm + (src_v ? (int64_t)k * NP : 0), src_v ? src_bytes : 0u);
#pragma unroll
    for (int r = 0; r < WD_WPT; ++r) {
      buf_load(fu[r], ru, gb[r], 0u);
      buf_load(fv[r], rv, gb[r], 0u);
      __builtin_amdgcn_sched_barrier(0);   // program order, here and in the loop: see below
    }
  };
  fetch(0);
  TD *uplane = (TD *)a.u, *vplane = (TD *)a.v;
  // gfx950 counts loads and stores in ONE in-order counter (vmcnt).  At the top of the level loop the compiler must wait for level k's
  // loads; what it may leave in flight is the minimum over the two ways into the loop of "memory operations issued after the load".
  // From the loop's own end those are the later loads and the level's four stores; from here, without the four (dropped: their lane
  // offset is out of range) stores below, nothing -- and the wait became vmcnt(0): every level waited for the stores of the level
  // before it to be acknowledged.  With the same six loads in the same order and four stores behind them on both ways in, the wait is
  // "the loads, not the stores" (vmcnt(5) / (3) / (1) in the ISA).  Measured level on configuration 4's grid (0.79-0.80 ms either way,
  // profiles/r06_wind_chain_probe.txt: the kernel runs at 91 % of the box's device-copy rate with two workgroups per CU); kept: it costs nothing.
  {
    const BufRsrc none = buf_rsrc(nullptr, 0u);
#pragma unroll
    for (int r = 0; r < 2 * WD_RPT; ++r) buf_store_lane((TD)0, none, nullptr, MPG_BUF_NONE, lane_bytes);
  }
  for (int k = 0; k < nlev; ++k) {
    double *bu = lds + (k & 1) * 2 * WD_NW, *bv = bu + WD_NW;
#pragma unroll
    for (int r = 0; r < WD_WPT; ++r) {
      const int e = t + WD_NT * r;
      double un = fu[r], vn = fv[r];
      if constexpr (ROT) wd_rotate(fu[r], fv[r], rca[r], rsa[r], rtn[r], rdn[r], un, vn);
      if (e < WD_NW) {
        bu[e] = un;
        bv[e] = vn;
      }
      if constexpr (KEEP) {
        if (a.um_rot && wown[r]) __builtin_nontemporal_store(un, a.um_rot + (int64_t)k * NP + (gb[r] >> 3));
        if (a.vm_rot && wown[r]) __builtin_nontemporal_store(vn, a.vm_rot + (int64_t)k * NP + (gb[r] >> 3));
      }
    }
    __syncthreads();
    fetch(min(k + 1, nlev - 1));   // no branch around the loads (the last level is fetched once more): the compiler can count what is in flight
    const BufRsrc ou = buf_rsrc(do_u ? uplane : nullptr, do_u ? u_bytes : 0u), ov = buf_rsrc(do_v ? vplane : nullptr, do_v ? v_bytes : 0u);
#pragma unroll
    for (int r = 0; r < WD_RPT; ++r) {
      {
        const double *q = bu + lu[r];
        double acc = fma(wu[r][0], q[0], 0.0);
        acc = fma(wu[r][1], q[1], acc);
        acc = fma(wu[r][2], q[WD_WW + 1], acc);
        acc = fma(wu[r][3], q[WD_WW], acc);
        buf_store_lane(finish(acc, mapu[r]), ou, uplane, pu[r], lane_bytes);
      }
      {
        const double *q = bv + lv[r];
        double acc = fma(wv[r][0], q[0], 0.0);
        acc = fma(wv[r][1], q[1], acc);
        acc = fma(wv[r][2], q[WD_WW + 1], acc);
        acc = fma(wv[r][3], q[WD_WW], acc);
        buf_store_lane(finish(acc, mapv[r]), ov, vplane, pv[r], lane_bytes);
      }
    }
    uplane += P1;
    vplane += P2;
  }

  // ---- points with a source outside the window: straight from global memory ---------------------------------
  bool any_far = false;
#pragma unroll
  for (int r = 0; r < WD_RPT; ++r) any_far = any_far || faru[r] || farv[r];
  if (!any_far) return;
  auto far_point = [&](const int32_t *idx, int64_t P, int64_t p, const double *w4, bool want_v, TD *out) {
    int32_t c[4];
    double ca[4], sa[4], tn[4], dn[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      c[q] = idx[q * P + p];
      ca[q] = 1.0; sa[q] = 0.0; tn[q] = 0.0; dn[q] = 1.0;
      if constexpr (ROT) {
        ca[q] = a.cosa[c[q]];
        sa[q] = a.sina[c[q]];
        tn[q] = sa[q] / ca[q];
        dn[q] = ca[q] + sa[q] * tn[q];
      }
    }
    for (int k = 0; k < nlev; ++k) {
      double acc = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double uo = (ROT || !want_v) ? a.um[(int64_t)k * NP + c[q]] : 0.0;
        const double vo = (ROT || want_v) ? a.vm[(int64_t)k * NP + c[q]] : 0.0;
        double un = uo, vn = vo;
        if constexpr (ROT) wd_rotate(uo, vo, ca[q], sa[q], tn[q], dn[q], un, vn);
        acc = fma(w4[q], want_v ? vn : un, acc);
      }
      out[(int64_t)k * P + p] = finish(acc, true);
    }
  };
#pragma unroll
  for (int r = 0; r < WD_RPT; ++r) {
    const int j = j0 + wave + (WD_NT / 64) * r;
    if (faru[r]) far_point(a.idx1, P1, (int64_t)j * nxu + (i0 + lane - (int)(((long long)j * nxu) % WD_A)), wu[r], false, (TD *)a.u);
    if (farv[r]) far_point(a.idx2, P2, (int64_t)j * nx + (i0 + lane - (int)(((long long)j * nx) % WD_A)), wv[r], true, (TD *)a.v);
  }
}

template <typename TD, bool TYPED>
static int launch_wind(const WindArgs &a, bool rot, unsigned nwg, hipStream_t s) {
  if (rot && (a.um_rot || a.vm_rot)) k_wind_destagger<TD, true, TYPED, true><<<nwg, WD_NT, 0, s>>>(a);
  else if (rot) k_wind_destagger<TD, true, TYPED, false><<<nwg, WD_NT, 0, s>>>(a);
  else k_wind_destagger<TD, false, TYPED, false><<<nwg, WD_NT, 0, s>>>(a);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

// h1 / h2: the CENTER -> EDGE1 / CENTER -> EDGE2 handles of ONE grid (either may be NULL: that component is not
// produced); cosa / sina NULL: no rotation.  -> MPG_ERR_UNSUPPORTED when the handles are not such a pair (the caller
// keeps the three-call chain).
int mpg_k_wind_destagger(mpg_handle_s *h1, mpg_handle_s *h2, const double *cosa, const double *sina, const double *um, const double *vm, int nlev,
                         void *u, void *v, int dst_type, double *um_rot, double *vm_rot, hipStream_t s) {
  int nx, ny;
  if (h2) {
    nx = h2->nx_dst;
    ny = h2->ny_dst - 1;
  } else {
    nx = h1->nx_dst - 1;
    ny = h1->ny_dst;
  }
  const int64_t NP = (int64_t)nx * ny;
  for (mpg_handle_s *h : {h1, h2}) {
    if (!h) continue;
    const bool e1 = h == h1;
    if (h->kind != MPG_KIND_FIXED || h->nnz_per_row != 4 || h->localized || h->n_src != NP || nx < 1 || ny < 1 || h->nx_dst != nx + (e1 ? 1 : 0) ||
        h->ny_dst != ny + (e1 ? 0 : 1))
      return MPG_ERR_UNSUPPORTED;
  }
  if ((int64_t)(nx + 1) * (ny + 1) * 8 >= 0xFFFFFFFFLL) return MPG_ERR_UNSUPPORTED;   // 32-bit byte offsets inside a level plane
  const bool rot = cosa != nullptr;
  const int64_t n_pole = (h1 ? h1->n_pole : 0) + (h2 ? h2->n_pole : 0);
  if (rot && n_pole) return MPG_ERR_UNSUPPORTED;   // (a rotated field under pole caps: no projection of the reference asks for it)
  if (nlev == 0) return MPG_SUCCESS;
  WindArgs a;
  a.idx1 = h1 ? h1->idx.p : nullptr;
  a.w1 = h1 ? h1->w.p : nullptr;
  a.idx2 = h2 ? h2->idx.p : nullptr;
  a.w2 = h2 ? h2->w.p : nullptr;
  a.cosa = cosa;
  a.sina = sina;
  a.um = um;
  a.vm = vm;
  a.u = u;
  a.v = v;
  a.um_rot = um_rot;
  a.vm_rot = vm_rot;
  a.nx = nx;
  a.ny = ny;
  a.nlev = nlev;
  a.periodic = ((h1 && h1->n_pole) || (h2 && h2->n_pole)) ? 1 : 0;
  a.mode = (h1 ? 1 : 0) | (h2 ? 2 : 0);
  a.ntx = (nx + 1 + WD_A - 1 + WD_TX - 1) / WD_TX;
  a.scale = 1.0;
  a.offset = 0.0;
  a.dbe = (dst_type & MPG_TYPE_BE) != 0;
  const int nty = (ny + 1 + WD_TY - 1) / WD_TY;
  const unsigned nwg = (unsigned)a.ntx * (unsigned)nty;
  const bool typed = dst_type != MPG_TYPE_F64;
  int rc;
  if (!typed) rc = launch_wind<double, false>(a, rot, nwg, s);
  else if (dst_type & MPG_TYPE_F32) rc = launch_wind<float, true>(a, rot, nwg, s);
  else rc = launch_wind<double, true>(a, rot, nwg, s);
  if (rc) return rc;
  // pole caps of a periodic grid: rewritten from the (unrotated) mass field exactly as after k_applyN
  if (h1 && h1->n_pole && (rc = mpg_k_pole_fix(h1, um, MPG_TYPE_F64, MPG_LAYOUT_CELL_FAST, nlev, 1, u, dst_type, 1.0, 0.0, s))) return rc;
  if (h2 && h2->n_pole && (rc = mpg_k_pole_fix(h2, vm, MPG_TYPE_F64, MPG_LAYOUT_CELL_FAST, nlev, 1, v, dst_type, 1.0, 0.0, s))) return rc;
  return MPG_SUCCESS;
}

// mpg_init loads this translation unit's code object ahead of its first launch (mpg_api.hip: warm_modules)
const void *mpg_anchor_k_wind() { return (const void *)&k_wind_destagger<double, true, false, false>; }
