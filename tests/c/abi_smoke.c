/* Plain-C consumer of include/mpassit_amd.h (compiled with gcc -std=c99, no C++, no Python, no torch):
 * proves the boundary is a real C-ABI.  Mesh = the 4-cell "tetrahedral" Voronoi diagram of the sphere
 * (cells at the tetrahedron vertices, Voronoi vertices at their antipodes, 3 cells per vertex), target = a
 * 12 x 6 global lat-lon grid.  Checks: constants are reproduced by bilinear and conservative regridding, nearest
 * returns one of the 4 source values bit for bit, the handle cache returns the same handle, errors are reported; the
 * device-resident typed Regrid takes and produces big-endian float32 (the bytes of a NetCDF classic variable); the source
 * range of a handle and the mesh's source window; a one-rank communicator, halo schedule, exchange and row gather (RCCL). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mpassit_amd.h"

#define CHECK(call)                                                                   \
  do {                                                                                \
    int rc_ = (call);                                                                 \
    if (rc_ != MPG_SUCCESS) {                                                         \
      fprintf(stderr, "FAIL %s -> %d: %s\n", #call, rc_, mpg_last_error());         \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

int main(void) {
  const double PI = 3.14159265358979323846;
  /* tetrahedron vertices (unit vectors) -> lat/lon in radians, lon in [0, 2pi) */
  const double t[4][3] = {{1, 1, 1}, {1, -1, -1}, {-1, 1, -1}, {-1, -1, 1}};
  double latC[4], lonC[4], latV[4], lonV[4];
  for (int i = 0; i < 4; ++i) {
    double n = sqrt(3.0), x = t[i][0] / n, y = t[i][1] / n, z = t[i][2] / n;
    latC[i] = asin(z);
    lonC[i] = atan2(y, x);
    if (lonC[i] < 0) lonC[i] += 2 * PI;
    latV[i] = asin(-z); /* Voronoi vertex i = antipode of cell i: equidistant from the other three cells */
    lonV[i] = atan2(-y, -x);
    if (lonV[i] < 0) lonV[i] += 2 * PI;
  }
  /* verticesOnCell [nCells][maxEdges], 1-based: cell c is bounded by the vertices j != c */
  int32_t voc[4][3] = {{2, 3, 4}, {1, 4, 3}, {1, 2, 4}, {1, 3, 2}};
  if (mpg_regrid_store(NULL, 0, NULL, 0, 0, NULL) != MPG_ERR_NOT_INITIALIZED) {
    fprintf(stderr, "FAIL: calls before mpg_init must return MPG_ERR_NOT_INITIALIZED\n");
    return 1;
  }
  int ngpu = -1;
  CHECK(mpg_device_count(&ngpu));   /* before mpg_init: a launcher's ranks choose their device with it */
  if (ngpu < 0) {
    fprintf(stderr, "FAIL: mpg_device_count left %d\n", ngpu);
    return 1;
  }
  CHECK(mpg_init(0));               /* (no GPU: fails here, saying that there is no CPU fallback) */
  if (ngpu < 1) {
    fprintf(stderr, "FAIL: mpg_init succeeded although mpg_device_count saw no GPU\n");
    return 1;
  }
  mpg_mesh mesh;
  CHECK(mpg_mesh_create(4, 4, 3, latC, lonC, latV, lonV, &voc[0][0], &mesh));
  enum { NX = 12, NY = 6 };
  double lon[NY][NX], lat[NY][NX], lonc[NY + 1][NX + 1], latc[NY + 1][NX + 1];
  for (int j = 0; j <= NY; ++j)
    for (int i = 0; i <= NX; ++i) {
      lonc[j][i] = -180.0 + 30.0 * i;
      latc[j][i] = -90.0 + 30.0 * j;
      if (i < NX && j < NY) {
        lon[j][i] = -165.0 + 30.0 * i;
        lat[j][i] = -75.0 + 30.0 * j;
      }
    }
  mpg_grid grid;
  CHECK(mpg_grid_create(NX, NY, 1, &lon[0][0], &lat[0][0], &lonc[0][0], &latc[0][0], NULL, NULL, NULL, NULL, &grid));
  const double src[2][4] = {{7.5, 7.5, 7.5, 7.5}, {1.0, 2.0, 3.0, 4.0}}; /* 2 levels, cell-fastest */
  double dst[2][NY][NX];
  int methods[3] = {MPG_REGRIDMETHOD_BILINEAR, MPG_REGRIDMETHOD_CONSERVE, MPG_REGRIDMETHOD_NEAREST_STOD};
  for (int m = 0; m < 3; ++m) {
    mpg_handle rh, rh2;
    CHECK(mpg_regrid_store(mesh, MPG_MESHLOC_ELEMENT, grid, MPG_STAGGERLOC_CENTER, methods[m], &rh));
    CHECK(mpg_regrid_store(mesh, MPG_MESHLOC_ELEMENT, grid, MPG_STAGGERLOC_CENTER, methods[m], &rh2));
    if (rh != rh2) {
      fprintf(stderr, "FAIL: handle cache\n");
      return 1;
    }
    CHECK(mpg_handle_release(rh2));
    memset(dst, 0xff, sizeof dst);
    CHECK(mpg_regrid(rh, &src[0][0], MPG_LAYOUT_CELL_FAST, 2, 1, &dst[0][0][0]));
    for (int j = 0; j < NY; ++j)
      for (int i = 0; i < NX; ++i) {
        if (fabs(dst[0][j][i] - 7.5) > 1e-9) {
          fprintf(stderr, "FAIL method %d: constant not reproduced at (%d,%d): %.17g\n", methods[m], i, j, dst[0][j][i]);
          return 1;
        }
        double v = dst[1][j][i];
        if (methods[m] == MPG_REGRIDMETHOD_NEAREST_STOD) {
          if (v != 1.0 && v != 2.0 && v != 3.0 && v != 4.0) {
            fprintf(stderr, "FAIL nearest: %.17g is not a source value\n", v);
            return 1;
          }
        } else if (!(v >= 1.0 - 1e-9 && v <= 4.0 + 1e-9)) {
          fprintf(stderr, "FAIL method %d: %.17g outside the convex hull of the sources\n", methods[m], v);
          return 1;
        }
      }
    CHECK(mpg_handle_release(rh));
  }
  /* ---- fused ingest / egress on device buffers: big-endian float32 in, big-endian float32 out (T - 300 fused) ---- */
  {
    mpg_handle rh;
    CHECK(mpg_regrid_store(mesh, MPG_MESHLOC_ELEMENT, grid, MPG_STAGGERLOC_CENTER, MPG_REGRIDMETHOD_NEAREST_STOD, &rh));
    const float f32[4] = {301.0f, 302.0f, 303.0f, 304.0f};
    unsigned char be_in[16], be_out[NY * NX * 4];
    for (int i = 0; i < 4; ++i) {
      unsigned char raw[4];
      memcpy(raw, &f32[i], 4);
      for (int b = 0; b < 4; ++b) be_in[4 * i + b] = raw[3 - b];   /* as a classic NetCDF file stores it (this host is little-endian) */
    }
    void *s_dev, *d_dev;
    CHECK(mpg_dev_alloc(16, &s_dev));
    CHECK(mpg_dev_alloc(NY * NX * 4, &d_dev));
    CHECK(mpg_dev_upload(s_dev, be_in, 16));
    CHECK(mpg_regrid_typed_dev(rh, s_dev, MPG_TYPE_F32 | MPG_TYPE_BE, MPG_LAYOUT_CELL_FAST, 1, 1, d_dev, MPG_TYPE_F32 | MPG_TYPE_BE, 1.0, -300.0, NULL));
    CHECK(mpg_dev_download(be_out, d_dev, NY * NX * 4));
    for (int p = 0; p < NY * NX; ++p) {
      unsigned char raw[4];
      float v;
      for (int b = 0; b < 4; ++b) raw[b] = be_out[4 * p + 3 - b];
      memcpy(&v, raw, 4);
      if (v != 1.0f && v != 2.0f && v != 3.0f && v != 4.0f) {
        fprintf(stderr, "FAIL big-endian typed Regrid: point %d = %.9g\n", p, v);
        return 1;
      }
    }
    /* ---- the same field twice as a bundle of separate arrays, one epilogue offset each (ESMF_FieldBundleRegrid) ---- */
    {
      void *d2_dev;
      unsigned char be2[NY * NX * 4];
      CHECK(mpg_dev_alloc(NY * NX * 4, &d2_dev));
      const void *srcs[2] = {s_dev, s_dev};
      void *dsts[2] = {d_dev, d2_dev};
      const double offs[2] = {-300.0, -299.0};
      CHECK(mpg_regrid_bundle_typed_dev(rh, 2, srcs, MPG_TYPE_F32 | MPG_TYPE_BE, MPG_LAYOUT_CELL_FAST, 1, dsts, MPG_TYPE_F32 | MPG_TYPE_BE, 1.0, offs, NULL));
      CHECK(mpg_dev_download(be2, d_dev, NY * NX * 4));
      if (memcmp(be2, be_out, NY * NX * 4) != 0) {
        fprintf(stderr, "FAIL bundle Regrid: field 0 differs from the single call\n");
        return 1;
      }
      CHECK(mpg_dev_download(be2, d2_dev, NY * NX * 4));
      for (int p = 0; p < NY * NX; ++p) {
        unsigned char raw[4];
        float v, v0;
        for (int b = 0; b < 4; ++b) raw[b] = be2[4 * p + 3 - b];
        memcpy(&v, raw, 4);
        for (int b = 0; b < 4; ++b) raw[b] = be_out[4 * p + 3 - b];
        memcpy(&v0, raw, 4);
        if (v != v0 + 1.0f) {
          fprintf(stderr, "FAIL bundle Regrid: point %d = %.9g, expected %.9g\n", p, v, v0 + 1.0f);
          return 1;
        }
      }
      CHECK(mpg_dev_free(d2_dev));
    }
    /* ---- ... and as a bundle of separate HOST arrays through one pipeline ---- */
    {
      float h0[NY * NX], h1[NY * NX];
      const void *hs[2] = {f32, f32};
      void *hd[2] = {h0, h1};
      const double hoffs[2] = {0.0, 10.0};
      CHECK(mpg_regrid_bundle_typed(rh, 2, hs, MPG_TYPE_F32, MPG_LAYOUT_CELL_FAST, 1, hd, MPG_TYPE_F32, 1.0, hoffs));
      for (int p = 0; p < NY * NX; ++p)
        if (h0[p] < 301.0f || h0[p] > 304.0f || h1[p] != h0[p] + 10.0f) {
          fprintf(stderr, "FAIL host bundle Regrid: point %d = %.9g / %.9g\n", p, h0[p], h1[p]);
          return 1;
        }
    }
    /* ---- source range and source window: all four cells are referenced; the whole mesh is the only window that fits ---- */
    int64_t first = -1, end = -1;
    CHECK(mpg_handle_source_range(rh, &first, &end));
    if (first != 0 || end != 4) {
      fprintf(stderr, "FAIL source range [%lld, %lld)\n", (long long)first, (long long)end);
      return 1;
    }
    if (mpg_mesh_set_source_window(mesh, MPG_MESHLOC_ELEMENT, 1, 3) == MPG_SUCCESS) {
      fprintf(stderr, "FAIL: a window that cuts into a handle must be refused\n");
      return 1;
    }
    CHECK(mpg_mesh_set_source_window(mesh, MPG_MESHLOC_ELEMENT, 0, 4));
    /* ---- round 6: Stores begun in the background, and interp.F90:291-328 (rotate_winds_cgrid + UMASS -> U + VMASS -> V) in one pass ---- */
    {
      enum { WX = 40, WY = 24, WL = 3 };
      mpg_proj pj;
      memset(&pj, 0, sizeof pj);
      pj.code = MPG_PROJ_LC;
      pj.known_lat = 38.5; pj.known_lon = -97.5; pj.known_x = 0.5 * (WX + 1); pj.known_y = 0.5 * (WY + 1);
      pj.dx_m = 30000.0; pj.stand_lon = -97.5; pj.truelat1 = 38.5; pj.truelat2 = 38.5;
      mpg_grid wg;
      CHECK(mpg_grid_create_proj(&pj, WX, WY, 0, &wg));
      CHECK(mpg_regrid_store_grid_begin(wg, MPG_STAGGERLOC_CENTER, MPG_STAGGERLOC_EDGE1, MPG_REGRIDMETHOD_BILINEAR));
      CHECK(mpg_regrid_store_grid_begin(wg, MPG_STAGGERLOC_CENTER, MPG_STAGGERLOC_EDGE2, MPG_REGRIDMETHOD_BILINEAR));
      double um[WL][WY][WX], vm[WL][WY][WX];
      for (int k = 0; k < WL; ++k)
        for (int j = 0; j < WY; ++j)
          for (int i = 0; i < WX; ++i) {
            um[k][j][i] = 10.0 + k + 0.3 * i - 0.2 * j;
            vm[k][j][i] = -4.0 + 0.5 * k + 0.1 * i * j;
          }
      mpg_handle ru, rv;
      CHECK(mpg_regrid_store_grid(wg, MPG_STAGGERLOC_CENTER, MPG_STAGGERLOC_EDGE1, MPG_REGRIDMETHOD_BILINEAR, &ru));   /* collects the begun Stores */
      CHECK(mpg_regrid_store_grid(wg, MPG_STAGGERLOC_CENTER, MPG_STAGGERLOC_EDGE2, MPG_REGRIDMETHOD_BILINEAR, &rv));
      const double *ca, *sa;
      CHECK(mpg_grid_rotang_dev(wg, &ca, &sa));
      const size_t nm = sizeof um, nu = (size_t)WL * WY * (WX + 1) * 8, nv = (size_t)WL * (WY + 1) * WX * 8;
      void *um_d, *vm_d, *ur_d, *vr_d, *u1_d, *v1_d, *u3_d, *v3_d;
      CHECK(mpg_dev_alloc(nm, &um_d)); CHECK(mpg_dev_alloc(nm, &vm_d)); CHECK(mpg_dev_alloc(nm, &ur_d)); CHECK(mpg_dev_alloc(nm, &vr_d));
      CHECK(mpg_dev_alloc(nu, &u1_d)); CHECK(mpg_dev_alloc(nv, &v1_d)); CHECK(mpg_dev_alloc(nu, &u3_d)); CHECK(mpg_dev_alloc(nv, &v3_d));
      CHECK(mpg_dev_upload(um_d, um, nm)); CHECK(mpg_dev_upload(vm_d, vm, nm));
      CHECK(mpg_dev_upload(ur_d, um, nm)); CHECK(mpg_dev_upload(vr_d, vm, nm));
      /* one pass ... */
      CHECK(mpg_wind_destagger_dev(ru, rv, ca, sa, (const double *)um_d, (const double *)vm_d, WL, u1_d, v1_d, MPG_TYPE_F64, NULL, NULL, NULL));
      /* ... against the three calls of the reference's sequence */
      CHECK(mpg_rotate_winds_dev((int64_t)WX * WY, WL, ca, sa, (double *)ur_d, (double *)vr_d, NULL));
      CHECK(mpg_regrid_dev(ru, (const double *)ur_d, MPG_LAYOUT_CELL_FAST, WL, 1, (double *)u3_d, NULL));
      CHECK(mpg_regrid_dev(rv, (const double *)vr_d, MPG_LAYOUT_CELL_FAST, WL, 1, (double *)v3_d, NULL));
      static double u1[WL][WY][WX + 1], u3[WL][WY][WX + 1], v1[WL][WY + 1][WX], v3[WL][WY + 1][WX];
      CHECK(mpg_dev_download(u1, u1_d, nu)); CHECK(mpg_dev_download(u3, u3_d, nu));
      CHECK(mpg_dev_download(v1, v1_d, nv)); CHECK(mpg_dev_download(v3, v3_d, nv));
      if (memcmp(u1, u3, nu) != 0 || memcmp(v1, v3, nv) != 0) {
        fprintf(stderr, "FAIL: mpg_wind_destagger_dev differs from rotate_winds + two Regrids\n");
        return 1;
      }
      if (u1[1][WY / 2][WX / 2] == 0.0 || u1[0][3][0] != 0.0 || v1[2][0][5] != 0.0) {   /* interior mapped, outer half-cell ring 0.0 */
        fprintf(stderr, "FAIL: staggered winds: interior / hull values\n");
        return 1;
      }
      if (mpg_wind_destagger_dev(ru, rv, ca, sa, (const double *)um_d, (const double *)vm_d, WL, u1_d, v1_d, MPG_TYPE_F64, (double *)um_d, NULL, NULL) !=
          MPG_ERR_INVALID_ARG) {
        fprintf(stderr, "FAIL: rotated mass winds over the inputs must be refused\n");
        return 1;
      }
      /* the same chain on HOST arrays (the reference's own shape): one call, the rotated mass winds back into the arrays that held them */
      {
        static double cah[WY][WX], sah[WY][WX], uh[WL][WY][WX + 1], vh[WL][WY + 1][WX], umh[WL][WY][WX], vmh[WL][WY][WX], urh[WL][WY][WX];
        CHECK(mpg_grid_get_rotang(wg, &cah[0][0], &sah[0][0]));
        memcpy(umh, um, nm);
        memcpy(vmh, vm, nm);
        CHECK(mpg_wind_destagger(ru, rv, &cah[0][0], &sah[0][0], &umh[0][0][0], &vmh[0][0][0], WL, uh, vh, MPG_TYPE_F64, &umh[0][0][0], &vmh[0][0][0]));
        CHECK(mpg_dev_download(urh, ur_d, nm));   /* what mpg_rotate_winds_dev left in place above */
        if (memcmp(uh, u3, nu) != 0 || memcmp(vh, v3, nv) != 0 || memcmp(umh, urh, nm) != 0) {
          fprintf(stderr, "FAIL: mpg_wind_destagger (host arrays) differs from rotate_winds + two Regrids\n");
          return 1;
        }
      }
      CHECK(mpg_handle_release(ru)); CHECK(mpg_handle_release(rv));
      CHECK(mpg_dev_free(um_d)); CHECK(mpg_dev_free(vm_d)); CHECK(mpg_dev_free(ur_d)); CHECK(mpg_dev_free(vr_d));
      CHECK(mpg_dev_free(u1_d)); CHECK(mpg_dev_free(v1_d)); CHECK(mpg_dev_free(u3_d)); CHECK(mpg_dev_free(v3_d));
      CHECK(mpg_grid_destroy(wg));
      CHECK(mpg_regrid_store_begin(mesh, MPG_MESHLOC_ELEMENT, grid, MPG_STAGGERLOC_CENTER, MPG_REGRIDMETHOD_CONSERVE));   /* begun, collected by nobody: parked */
    }
    /* ---- one rank of the multi-GPU verbs: communicator, halo schedule of this handle, exchange, row gather ---- */
    mpg_comm comm;
    mpg_halo halo;
    CHECK(mpg_comm_init(0, 1, NULL, &comm));
    CHECK(mpg_handle_release(rh));                 /* parked in the cache; a fresh Store below must hand it out again */
    CHECK(mpg_regrid_store(mesh, MPG_MESHLOC_ELEMENT, grid, MPG_STAGGERLOC_CENTER, MPG_REGRIDMETHOD_NEAREST_STOD, &rh));
    CHECK(mpg_halo_build(comm, rh, 4, 0, &halo));
    int mode;
    int64_t n_local, own[2], base, own_pos[2], sent, received;
    CHECK(mpg_halo_info(halo, &mode, &n_local, own, &base, own_pos, &sent, &received));
    if (n_local != 4 || own[0] != 0 || own[1] != 4 || sent != 0 || received != 0) {
      fprintf(stderr, "FAIL halo of one rank: n_local %lld own [%lld, %lld) sent %lld\n", (long long)n_local, (long long)own[0], (long long)own[1],
              (long long)sent);
      return 1;
    }
    void *src_dev, *loc_dev, *dst_dev, *all_dev;
    CHECK(mpg_dev_alloc(2 * 4 * 8, &src_dev));
    CHECK(mpg_dev_alloc(2 * 4 * 8, &loc_dev));
    CHECK(mpg_dev_alloc(2 * NY * NX * 8, &dst_dev));
    CHECK(mpg_dev_alloc(2 * NY * NX * 8, &all_dev));
    CHECK(mpg_dev_upload(src_dev, &src[0][0], 2 * 4 * 8));
    /* range form: the own block sits in the local buffer at own_pos[0]; here the local space IS the own block */
    CHECK(mpg_dev_upload((char *)loc_dev + own_pos[0] * 8, &src[0][0], 4 * 8));
    CHECK(mpg_dev_upload((char *)loc_dev + (4 + own_pos[0]) * 8, &src[1][0], 4 * 8));
    CHECK(mpg_halo_exchange_dev(halo, (char *)loc_dev + own_pos[0] * 8, n_local, loc_dev, 2, 8, NULL));
    CHECK(mpg_regrid_dev(rh, (const double *)loc_dev, MPG_LAYOUT_CELL_FAST, 2, 1, (double *)dst_dev, NULL));
    CHECK(mpg_gather_rows(comm, dst_dev, 0, NY, NX, NY, 2, 8, all_dev, 0, NULL));
    CHECK(mpg_dev_download(&dst[0][0][0], all_dev, 2 * NY * NX * 8));
    for (int j = 0; j < NY; ++j)
      for (int i = 0; i < NX; ++i)
        if (dst[0][j][i] != 7.5 || (dst[1][j][i] != 1.0 && dst[1][j][i] != 2.0 && dst[1][j][i] != 3.0 && dst[1][j][i] != 4.0)) {
          fprintf(stderr, "FAIL halo + gather path at (%d,%d): %.17g %.17g\n", i, j, dst[0][j][i], dst[1][j][i]);
          return 1;
        }
    CHECK(mpg_halo_destroy(halo));
    CHECK(mpg_comm_destroy(comm));
    CHECK(mpg_handle_release(rh));
    CHECK(mpg_dev_free(s_dev)); CHECK(mpg_dev_free(d_dev)); CHECK(mpg_dev_free(src_dev)); CHECK(mpg_dev_free(loc_dev));
    CHECK(mpg_dev_free(dst_dev)); CHECK(mpg_dev_free(all_dev));
  }
  /* ---- round-5 verbs: the caller's own partition of the cells (owned halo form), Store statistics, mpg_warmup_wait ---- */
  {
    mpg_comm comm;
    mpg_halo halo;
    mpg_handle rh;
    CHECK(mpg_warmup_wait());
    CHECK(mpg_comm_init(0, 1, NULL, &comm));
    CHECK(mpg_regrid_store(mesh, MPG_MESHLOC_ELEMENT, grid, MPG_STAGGERLOC_CENTER, MPG_REGRIDMETHOD_BILINEAR, &rh));
    int64_t st[8];
    CHECK(mpg_handle_store_stats(rh, st, 8));
    if (st[0] < 0 || st[2] <= 0) {   /* bilinear: [2] = triangles in all */
      fprintf(stderr, "FAIL store stats %lld %lld %lld\n", (long long)st[0], (long long)st[1], (long long)st[2]);
      return 1;
    }
    const int32_t twice[2] = {1, 1}, all4[4] = {0, 1, 2, 3}, three[3] = {0, 1, 2};
    if (mpg_halo_build_owned(comm, rh, 4, twice, 2, &halo) != MPG_ERR_INVALID_ARG ||      /* not sorted / unique */
        mpg_halo_build_owned(comm, rh, 4, three, 3, &halo) != MPG_ERR_INVALID_ARG) {      /* cell 3 is referenced and nobody owns it */
      fprintf(stderr, "FAIL: a partition that is none must be refused\n");
      return 1;
    }
    CHECK(mpg_halo_build_owned(comm, rh, 4, all4, 4, &halo));
    int mode;
    int64_t n_local, own[2], base, own_pos[2], sent, received;
    CHECK(mpg_halo_info(halo, &mode, &n_local, own, &base, own_pos, &sent, &received));
    if (mode != 2 || n_local != 4 || own[0] != 0 || own[1] != 4 || sent != 0 || received != 0) {
      fprintf(stderr, "FAIL owned halo of one rank: mode %d n_local %lld own [%lld, %lld)\n", mode, (long long)n_local, (long long)own[0], (long long)own[1]);
      return 1;
    }
    /* float32 sources in file order: ONE whole row of 2 levels is the exchanged element (nrows = 1 field, elem_bytes = 2 * 4) */
    const float rows32[4][2] = {{1.f, 10.f}, {2.f, 20.f}, {3.f, 30.f}, {4.f, 40.f}};
    float out32[2][NY][NX];
    void *own_dev, *loc_dev, *dst_dev;
    CHECK(mpg_dev_alloc(sizeof rows32, &own_dev));
    CHECK(mpg_dev_alloc(sizeof rows32, &loc_dev));
    CHECK(mpg_dev_alloc(sizeof out32, &dst_dev));
    CHECK(mpg_dev_upload(own_dev, rows32, sizeof rows32));
    CHECK(mpg_halo_exchange_dev(halo, own_dev, 4, loc_dev, 1, 2 * 4, NULL));
    CHECK(mpg_regrid_typed_dev(rh, loc_dev, MPG_TYPE_F32, MPG_LAYOUT_LEV_FAST, 2, 1, dst_dev, MPG_TYPE_F32, 1.0, 0.0, NULL));
    CHECK(mpg_dev_download(out32, dst_dev, sizeof out32));
    for (int j = 0; j < NY; ++j)
      for (int i = 0; i < NX; ++i)
        if (!(out32[0][j][i] >= 1.f && out32[0][j][i] <= 4.f) || !(out32[1][j][i] >= 10.f && out32[1][j][i] <= 40.f) ||
            fabs((double)out32[1][j][i] - 10.0 * (double)out32[0][j][i]) > 1e-4) {      /* level 2 = 10 x level 1 on every source: so on every point */
          fprintf(stderr, "FAIL owned halo + file-order Regrid at (%d,%d): %g %g\n", i, j, out32[0][j][i], out32[1][j][i]);
          return 1;
        }
    CHECK(mpg_halo_destroy(halo));
    CHECK(mpg_comm_destroy(comm));
    CHECK(mpg_handle_release(rh));
    CHECK(mpg_dev_free(own_dev)); CHECK(mpg_dev_free(loc_dev)); CHECK(mpg_dev_free(dst_dev));
  }
  /* ---- round-4 verbs: a mesh cut to a grid gives the whole mesh's weights; a projection that does not fit a grid is refused ---- */
  {
    mpg_mesh cut;
    mpg_handle ra, rb;
    CHECK(mpg_mesh_create_window(4, 4, 3, latC, lonC, latV, lonV, &voc[0][0], grid, &cut));
    int64_t c0, cn, v0, vn;
    double margin;
    CHECK(mpg_mesh_window_info(cut, &c0, &cn, &v0, &vn, &margin));
    if (c0 != 0 || cn != 4 || vn != 4) {   /* a global grid sees all four cells of the tetrahedron */
      fprintf(stderr, "FAIL mesh window [%lld, +%lld) vertices +%lld\n", (long long)c0, (long long)cn, (long long)vn);
      return 1;
    }
    CHECK(mpg_regrid_store(mesh, MPG_MESHLOC_ELEMENT, grid, MPG_STAGGERLOC_CENTER, MPG_REGRIDMETHOD_BILINEAR, &ra));
    CHECK(mpg_regrid_store(cut, MPG_MESHLOC_ELEMENT, grid, MPG_STAGGERLOC_CENTER, MPG_REGRIDMETHOD_BILINEAR, &rb));
    int32_t ia[NY * NX * 3], ib[NY * NX * 3];
    double wa[NY * NX * 3], wb[NY * NX * 3];
    CHECK(mpg_handle_get_weights(ra, ia, wa));
    CHECK(mpg_handle_get_weights(rb, ib, wb));
    if (memcmp(ia, ib, sizeof ia) || memcmp(wa, wb, sizeof wa)) {
      fprintf(stderr, "FAIL: the mesh cut to the grid gives other weights than the whole mesh\n");
      return 1;
    }
    CHECK(mpg_handle_release(ra));
    CHECK(mpg_handle_release(rb));
    CHECK(mpg_mesh_destroy(cut));
    mpg_proj pr;
    memset(&pr, 0, sizeof pr);
    pr.code = MPG_PROJ_LATLON;
    pr.known_lat = -75.0; pr.known_lon = -165.0; pr.known_x = 1.0; pr.known_y = 1.0;
    pr.dlat_deg = 30.0; pr.dlon_deg = 30.0;
    CHECK(mpg_grid_attach_proj(grid, &pr, 0));            /* the grid's own projection (30-degree cells: accepted, the search stays on the pyramid) */
    pr.dlon_deg = 1.0; pr.dlat_deg = 1.0;                  /* a fine projection that does not reproduce this grid's points */
    if (mpg_grid_attach_proj(grid, &pr, 0) != MPG_ERR_INVALID_ARG || strlen(mpg_last_error()) == 0) {
      fprintf(stderr, "FAIL: a projection that does not fit the grid must be refused with a message\n");
      return 1;
    }
    if (MPG_ERR_TIMEOUT != 6) return 1;                    /* the multi-rank waits' error code is part of the ABI */
  }
  /* error contract: bad argument -> rc != 0 and a message */
  mpg_handle bad;
  if (mpg_regrid_store(mesh, MPG_MESHLOC_ELEMENT, grid, MPG_STAGGERLOC_EDGE1, MPG_REGRIDMETHOD_BILINEAR, &bad) == MPG_SUCCESS ||
      strlen(mpg_last_error()) == 0) {
    fprintf(stderr, "FAIL: EDGE1 without coordinates must be an error with a message\n");
    return 1;
  }
  CHECK(mpg_mesh_destroy(mesh));
  CHECK(mpg_grid_destroy(grid));
  CHECK(mpg_finalize());
  printf("abi_smoke ok\n");
  return 0;
}
