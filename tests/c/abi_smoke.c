/* Plain-C consumer of include/mpassit_amd.h (compiled with gcc -std=c99, no C++, no Python, no torch):
 * proves the boundary is a real C-ABI.  Mesh = the 4-cell "tetrahedral" Voronoi diagram of the sphere
 * (cells at the tetrahedron vertices, Voronoi vertices at their antipodes, 3 cells per vertex), target = a
 * 12 x 6 global lat-lon grid.  Checks: constants are reproduced by bilinear and conservative regridding, nearest
 * returns one of the 4 source values bit for bit, the handle cache returns the same handle, errors are reported. */
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
  CHECK(mpg_init(0));
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
