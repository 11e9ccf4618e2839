// The wind chain of interp_hist_data as ONE pass over the mass-point winds (SURVEY s2.2 K6 + K7):
//
//   rotate_winds_cgrid(u_target_grid_nostag, v_target_grid_nostag)            interp.F90:291-293, 689-749
//   ESMF_FieldRegridStore / Regrid  UMASS(CENTER) -> U(EDGE1)                 interp.F90:295-311
//   ESMF_FieldRegridStore / Regrid  VMASS(CENTER) -> V(EDGE2)                 interp.F90:313-328
//
// The reference (and this library until round 5: k_rotate + 2 x k_applyN<4>) makes three passes: the rotation reads and
// rewrites both mass fields, each destaggering reads one of them again.  u/v_target_grid_nostag are intermediates the
// reference never writes to its file (write_data.F90 has no `nostag`), so here the earth-relative mass winds are read ONCE:
// a workgroup owns 64 x 16 points of the (i, j) index space, stages the (64 + 2 + 15) x (16 + 2) window of mass points
// around them in LDS -- rotated on the way in, with the operation sequence of interp.F90:737-748 (no contraction, as
// k_rotate: bit-identical) -- and combines every U point (EDGE1, nx + 1 columns) and every V point (EDGE2, ny + 1 rows) of
// its tile from there with the 4-point weights of the two Grid -> Grid handles, in k_applyN's accumulation order (the same
// bits as the three-pass chain).  Algorithmic bytes per level: 2 x 8 B read + 2 x e_dst written per mass point (+ 2 x 48 B
// of indices and weights and 16 B of rotation angles per point, once per launch) against 4 x 8 + 2 x (8 + e_dst) before.
//
// Store segments: a wave stores 64 consecutive points of ONE output row, and each row's segment is shifted left so that it
// starts on a multiple of 16 elements in memory whatever the row length (U rows are nx + 1 = 1801 wide on the README's
// grid: unshifted, every segment would straddle one more line and leave two partial lines to another workgroup); the 15
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
#define WD_TY 16
#define WD_NT 512
#define WD_A 16                         // store segments start on multiples of 16 elements
#define WD_WW (WD_TX + 2 + WD_A - 1)    // 81 window columns: i0 - 16 .. i0 + 64
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

template <typename TD, bool ROT, bool TYPED>
__global__ __launch_bounds__(WD_NT) void k_wind_destagger(const WindArgs a) {
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

  // ---- this thread's window points ---------------------------------------------------------------------------
  int32_t goff[WD_WPT];
  bool wval[WD_WPT], wown[WD_WPT];
  double rca[WD_WPT], rsa[WD_WPT], rtn[WD_WPT], rdn[WD_WPT];
#pragma unroll
  for (int r = 0; r < WD_WPT; ++r) {
    const int e = t + WD_NT * r, ly = e / WD_WW, lx = e - ly * WD_WW;
    const int cj = wj0 + ly;
    int ci = wi0 + lx;
    const bool raw_in = ci >= 0 && ci < nx;
    if (per) ci = ((ci % nx) + nx) % nx;
    wval[r] = e < WD_NW && cj >= 0 && cj < ny && ci >= 0 && ci < nx;
    wown[r] = wval[r] && raw_in && lx >= WD_A && lx < WD_A + WD_TX && ly >= 1 && ly <= WD_TY;
    goff[r] = wval[r] ? cj * nx + ci : 0;
    rca[r] = 1.0; rsa[r] = 0.0; rtn[r] = 0.0; rdn[r] = 1.0;
    if constexpr (ROT) {
      if (wval[r]) {
        rca[r] = a.cosa[goff[r]];
        rsa[r] = a.sina[goff[r]];
        rtn[r] = rsa[r] / rca[r];
        rdn[r] = rca[r] + rsa[r] * rtn[r];
      }
    }
  }

  // ---- this thread's output points: WD_RPT rows of U and of V ---------------------------------------------------
  int lu[WD_RPT][4], lv[WD_RPT][4];
  double wu[WD_RPT][4], wv[WD_RPT][4];
  bool actu[WD_RPT], actv[WD_RPT], mapu[WD_RPT], mapv[WD_RPT], faru[WD_RPT], farv[WD_RPT];
  int64_t pu[WD_RPT], pv[WD_RPT];
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
#pragma unroll
  for (int r = 0; r < WD_RPT; ++r) {
    const int j = j0 + wave + (WD_NT / 64) * r;
    {
      const int i = i0 + lane - (int)(((long long)j * nxu) % WD_A);
      actu[r] = do_u && j < ny && i >= 0 && i < nxu;
      pu[r] = actu[r] ? (int64_t)j * nxu + i : 0;
      mapu[r] = false;
      faru[r] = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) { lu[r][q] = 0; wu[r][q] = 0.0; }
      if (actu[r]) {
        int32_t c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          c[q] = a.idx1[q * P1 + pu[r]];
          wu[r][q] = a.w1[q * P1 + pu[r]];
        }
        mapu[r] = c[0] >= 0;
        if (mapu[r]) {
#pragma unroll
          for (int q = 0; q < 4; ++q) lu[r][q] = locate(c[q], faru[r]);
        }
      }
    }
    {
      const int i = i0 + lane - (int)(((long long)j * nx) % WD_A);
      actv[r] = do_v && j <= ny && i >= 0 && i < nx;
      pv[r] = actv[r] ? (int64_t)j * nx + i : 0;
      mapv[r] = false;
      farv[r] = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) { lv[r][q] = 0; wv[r][q] = 0.0; }
      if (actv[r]) {
        int32_t c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          c[q] = a.idx2[q * P2 + pv[r]];
          wv[r][q] = a.w2[q * P2 + pv[r]];
        }
        mapv[r] = c[0] >= 0;
        if (mapv[r]) {
#pragma unroll
          for (int q = 0; q < 4; ++q) lv[r][q] = locate(c[q], farv[r]);
        }
      }
    }
  }

  auto emit = [&](double val, bool mapped, TD *plane, int64_t p) {
    val = mapped ? val : 0.0;
    if constexpr (TYPED) {
      val = fma(val, a.scale, a.offset);
      __builtin_nontemporal_store(swz<true>((TD)val, zd), plane + p);
    } else {
      __builtin_nontemporal_store((TD)val, plane + p);
    }
  };

  // ---- level loop -------------------------------------------------------------------------------------------
  double fu[WD_WPT], fv[WD_WPT];
  auto fetch = [&](int k) {
    const double *uk = a.um + (int64_t)k * NP, *vk = a.vm + (int64_t)k * NP;
#pragma unroll
    for (int r = 0; r < WD_WPT; ++r) {
      fu[r] = (src_u && wval[r]) ? uk[goff[r]] : 0.0;
      fv[r] = (src_v && wval[r]) ? vk[goff[r]] : 0.0;
    }
  };
  fetch(0);
  TD *uplane = (TD *)a.u, *vplane = (TD *)a.v;
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
      if (wown[r]) {
        if (a.um_rot) __builtin_nontemporal_store(un, a.um_rot + (int64_t)k * NP + goff[r]);
        if (a.vm_rot) __builtin_nontemporal_store(vn, a.vm_rot + (int64_t)k * NP + goff[r]);
      }
    }
    __syncthreads();
    if (k + 1 < nlev) fetch(k + 1);
#pragma unroll
    for (int r = 0; r < WD_RPT; ++r) {
      if (actu[r] && !faru[r]) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = fma(wu[r][q], bu[lu[r][q]], acc);
        emit(acc, mapu[r], uplane, pu[r]);
      }
      if (actv[r] && !farv[r]) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = fma(wv[r][q], bv[lv[r][q]], acc);
        emit(acc, mapv[r], vplane, pv[r]);
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
  auto far_point = [&](const int32_t *idx, int64_t P, int64_t p, const double *w4, bool want_v, TD *out, int64_t plane) {
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
      emit(acc, true, out + (int64_t)k * plane, p);
    }
  };
#pragma unroll
  for (int r = 0; r < WD_RPT; ++r) {
    if (faru[r]) far_point(a.idx1, P1, pu[r], wu[r], false, (TD *)a.u, P1);
    if (farv[r]) far_point(a.idx2, P2, pv[r], wv[r], true, (TD *)a.v, P2);
  }
}

template <typename TD, bool TYPED>
static int launch_wind(const WindArgs &a, bool rot, unsigned nwg, hipStream_t s) {
  if (rot) k_wind_destagger<TD, true, TYPED><<<nwg, WD_NT, 0, s>>>(a);
  else k_wind_destagger<TD, false, TYPED><<<nwg, WD_NT, 0, s>>>(a);
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
  if ((int64_t)(nx + 1) * (ny + 1) >= 0x7fffffffLL) return MPG_ERR_UNSUPPORTED;
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
const void *mpg_anchor_k_wind() { return (const void *)&k_wind_destagger<double, true, false>; }
