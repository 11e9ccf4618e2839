// Target grid straight from the projection.
//
// The reference fills four staggered lat/lon arrays, three map-factor arrays and cos/sin(alpha) with host loops
// (define_target_grid_params, model_grid.F90:736-1038) before the hot path can start; on a 1800 x 1060 grid the
// numpy mirror of those loops costs ~20x the whole interpolation job.  Here one kernel per stagger evaluates
//   get_lat_lon_fields (model_grid.F90:2188-2219): xytoll(real(i-0.5)+0.5, real(j-0.5)+0.5, stagger)
//   xytoll             (llxy_module.F90:166-216):   U: x-0.5, V: y-0.5, CORNER: both
//   ij_to_latlon       (module_map_utils.F90:629-679; Lambert :1160-1233, lat-lon :1398-1428, polar stereographic
//                       :763-822, Mercator :1344-1362)
// and writes lon/lat (degrees), the unit vector every Store kernel works with, and the map factor
// (get_map_factor, model_grid.F90:2229-2365); a second kernel does get_rotang (:2450-2507) on the CENTER points.
// Pure ALU + transcendental work, a few MB written once: nowhere near any roofline, it only has to be off the host.
// Floating point: contraction is switched off so the arithmetic is the reference's operation for operation; the
// device libm (atan2/pow/tan/...) may differ from the host's in the last bit, the parity test allows 1e-12 degrees.
// No floating-point contraction in this translation unit (see k_store_conserve.hip): what it computes -- weights, coordinates --
// is a function of the source text, not of which product the compiler chooses to fuse; explicit fma() calls stay what they are.
#pragma clang fp contract(off)
#include <math.h>
#include <string.h>

#include "geom.h"
#include "mpg_internal.h"

#define TG_PI 3.141592653589793  // constants_module.F90:8
#define TG_RAD_PER_DEG (TG_PI / 180.0)
#define TG_DEG_PER_RAD (180.0 / TG_PI)
#define TG_EARTH_RADIUS_M 6370000.0  // constants_module.F90:25

__device__ __forceinline__ void ij_to_latlon(const ProjDev &p, double i, double j, double *lat_out, double *lon_out) {
#pragma clang fp contract(off)
  if (p.code == MPG_PROJ_LC) {
    double chi1 = (90.0 - p.hemi * p.truelat1) * TG_RAD_PER_DEG;
    double chi2 = (90.0 - p.hemi * p.truelat2) * TG_RAD_PER_DEG;
    double xx = p.hemi * i - p.polei;
    double yy = p.polej - p.hemi * j;
    double r2 = xx * xx + yy * yy;
    double r = sqrt(r2) / p.rebydx;
    double lat, lon;
    if (r2 == 0.0) {
      lat = p.hemi * 90.0;
      lon = p.stdlon;
    } else {
      lon = p.stdlon + TG_DEG_PER_RAD * atan2(p.hemi * xx, yy) / p.cone;
      lon = fmod(lon + 360.0, 360.0);
      double chi;
      if (chi1 == chi2) chi = 2.0 * atan(pow(r / tan(chi1), 1.0 / p.cone) * tan(chi1 * 0.5));
      else chi = 2.0 * atan(pow(r * p.cone / sin(chi1), 1.0 / p.cone) * tan(chi1 * 0.5));
      lat = (90.0 - chi * TG_DEG_PER_RAD) * p.hemi;
    }
    if (lon > 180.0) lon -= 360.0;
    if (lon < -180.0) lon += 360.0;
    *lat_out = lat;
    *lon_out = lon;
  } else if (p.code == MPG_PROJ_PS) {                     // ijll_ps, module_map_utils.F90:763-822
    double reflon = p.stdlon + 90.0;
    double scale_top = 1.0 + p.hemi * sin(p.truelat1 * TG_RAD_PER_DEG);
    double xx = i - p.polei;
    double yy = (j - p.polej) * p.hemi;
    double r2 = xx * xx + yy * yy;
    double lat, lon;
    if (r2 == 0.0) {
      lat = p.hemi * 90.0;
      lon = reflon;
    } else {
      double gi = p.rebydx * scale_top;
      double gi2 = gi * gi;
      lat = TG_DEG_PER_RAD * p.hemi * asin((gi2 - r2) / (gi2 + r2));
      double c = xx / sqrt(r2);
      c = fmin(fmax(c, -1.0), 1.0);
      double arccos = acos(c);
      lon = yy > 0.0 ? reflon + TG_DEG_PER_RAD * arccos : reflon - TG_DEG_PER_RAD * arccos;
    }
    if (lon > 180.0) lon -= 360.0;
    if (lon < -180.0) lon += 360.0;
    *lat_out = lat;
    *lon_out = lon;
  } else if (p.code == MPG_PROJ_MERC) {                   // ijll_merc, module_map_utils.F90:1344-1362
    double lat = 2.0 * atan(exp(p.dlon * (p.rsw + j - p.knownj))) * TG_DEG_PER_RAD - 90.0;
    double lon = (i - p.knowni) * p.dlon * TG_DEG_PER_RAD + p.lon1;
    if (lon > 180.0) lon -= 360.0;
    if (lon < -180.0) lon += 360.0;
    *lat_out = lat;
    *lon_out = lon;
  } else {
    double span = (double)(p.nxmax - p.nxmin + 1);
    double iw = i;
    if (i < p.nxmin - 0.5) iw = i + span;
    if (i >= p.nxmax + 0.5) iw = i - span;
    *lat_out = p.lat1 + (j - p.knownj) * p.latinc;
    *lon_out = p.lon1 + (iw - p.knowni) * p.loninc;
  }
}

__host__ __device__ __forceinline__ double map_factor(const ProjDev &p, double lat) {
#pragma clang fp contract(off)
  if (p.code == MPG_PROJ_PS)       // model_grid.F90:2279-2286
    return (1.0 + sin(TG_RAD_PER_DEG * fabs(p.truelat1))) / (1.0 + sin(TG_RAD_PER_DEG * copysign(1.0, p.truelat1) * lat));
  if (p.code == MPG_PROJ_MERC) {   // :2289-2298
    double colat0 = TG_RAD_PER_DEG * (90.0 - p.truelat1);
    return sin(colat0) / sin(TG_RAD_PER_DEG * (90.0 - lat));
  }
  if (p.code != MPG_PROJ_LC) return 1.0;  // PROJ_LATLON: no branch in get_map_factor
  double colat = TG_RAD_PER_DEG * (90.0 - lat);
  if (p.truelat1 != p.truelat2) {
    double colat1 = TG_RAD_PER_DEG * (90.0 - p.truelat1), colat2 = TG_RAD_PER_DEG * (90.0 - p.truelat2);
    double n = (log(sin(colat1)) - log(sin(colat2))) / (log(tan(colat1 / 2.0)) - log(tan(colat2 / 2.0)));
    return sin(colat2) / sin(colat) * pow(tan(colat / 2.0) / tan(colat2 / 2.0), n);
  }
  double colat0 = TG_RAD_PER_DEG * (90.0 - p.truelat1);
  return sin(colat0) / sin(colat) * pow(tan(colat / 2.0) / tan(colat0 / 2.0), cos(colat0));
}

// ---- the inverse: where on the grid does a point of the sphere fall?  (round 4) --------------------------------------------
// latlon_to_ij for the four projections program_setup.F90:166-191 accepts (llij_lc, module_map_utils.F90:1236-1290; llij_latlon,
// :1365-1392; since round 5 llij_ps :718-760 and llij_merc :1320-1341), in the 0-based CENTER index space (projection coordinate - 1).  The Stores use it to find the handful of target
// points around a source triangle / polygon directly instead of descending ten levels of the box pyramid; it only has to be
// good to a fraction of a grid length (the callers pad their boxes and test every candidate point exactly as before).
// Not usable -> NaN: within 1 degree of the Lambert pole or beyond 60 degrees into the other hemisphere; poleward of
// latlon_limit degrees on a lat-lon grid (great circles bend too much in index space there).
__global__ __launch_bounds__(256) void k_points_ij(ProjDev p, int row0, double latlon_limit, double i_center, int64_t n,
                                                   const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                   float *__restrict__ ij) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= n) return;
  const double lat = asin(fmin(fmax(z[q], -1.0), 1.0)) * TG_DEG_PER_RAD, lon = atan2(y[q], x[q]) * TG_DEG_PER_RAD;
  double i = NAN, j = NAN;
  if (p.code == MPG_PROJ_LC) {
    if (p.hemi * lat < 89.0 && p.hemi * lat > -60.0) {
      double dl = lon - p.stdlon;
      if (dl > 180.0) dl -= 360.0;
      if (dl < -180.0) dl += 360.0;
      const double rm = p.rebydx * cos(p.truelat1 * TG_RAD_PER_DEG) / p.cone *
                        pow(tan((90.0 * p.hemi - lat) * TG_RAD_PER_DEG / 2.0) / tan((90.0 * p.hemi - p.truelat1) * TG_RAD_PER_DEG / 2.0), p.cone);
      const double arg = p.cone * (dl * TG_RAD_PER_DEG);
      i = p.hemi * (p.polei + p.hemi * rm * sin(arg));
      j = p.hemi * (p.polej - rm * cos(arg));
    }
  } else if (p.code == MPG_PROJ_LATLON) {
    if (fabs(lat) < latlon_limit) {
      i = (lon - p.lon1) / p.loninc + p.knowni;
      j = (lat - p.lat1) / p.latinc + p.knownj;
      if (i_center == i_center) {
        // a REGIONAL lat-lon grid asked for the index nearest to its own columns (the nearest search bins cells by it): the
        // reference's wrap below sends a longitude just west of the first column a whole circle east
        const double period = 360.0 / p.loninc;
        i -= nearbyint((i - i_center) / period) * period;
      } else {
        const double span = (double)(p.nxmax - p.nxmin + 1);
        if (i < p.nxmin - 0.5) i += span;
        if (i >= p.nxmax + 0.5) i -= span;
      }
    }
  } else if (p.code == MPG_PROJ_PS) {
    // llij_ps (module_map_utils.F90:718-760).  The projection's own pole is a regular point of this map (no exclusion there, no
    // cut: the plane is continuous around it); what is singular is the OTHER pole -- beyond 60 degrees into the other hemisphere the
    // map factor passes 7 and nothing is handed out
    if (p.hemi * lat > -60.0) {
      const double reflon = p.stdlon + 90.0;
      const double scale_top = 1.0 + p.hemi * sin(p.truelat1 * TG_RAD_PER_DEG);
      const double ala = lat * TG_RAD_PER_DEG;
      const double rm = p.rebydx * cos(ala) * scale_top / (1.0 + p.hemi * sin(ala));
      const double alo = (lon - reflon) * TG_RAD_PER_DEG;
      i = p.polei + rm * cos(alo);
      j = p.polej + p.hemi * rm * sin(alo);
    }
  } else if (p.code == MPG_PROJ_MERC) {
    // llij_merc (module_map_utils.F90:1320-1341), with the longitude difference taken on the branch nearest to the grid's own
    // middle column instead of nearest to the known point (i_center: always given for this projection): the map's cut then sits
    // opposite the grid, wherever its known point is.  Both poles are singular: nothing beyond 85 degrees.
    if (fabs(lat) < 85.0) {
      double dl = lon - p.lon1;
      i = p.knowni + dl / (p.dlon * TG_DEG_PER_RAD);
      const double period = 360.0 / (p.dlon * TG_DEG_PER_RAD);
      const double ic = i_center == i_center ? i_center : p.knowni;
      i -= nearbyint((i - ic) / period) * period;
      j = p.knownj + log(tan(0.5 * ((lat + 90.0) * TG_RAD_PER_DEG))) / p.dlon - p.rsw;
    }
  }
  ij[2 * q] = (float)(i - 1.0);
  ij[2 * q + 1] = (float)(j - 1.0 - (double)row0);
}

// The index-space search rests on small-angle geometry: a figure a few index units across must span a few degrees at most, and a
// destination cell's own great-circle edges must stay inside its index band (the edge between two lat-lon corners 90 degrees
// of longitude apart rises by several degrees in between).  Grids coarser than 2 degrees / 200 km per cell, or whose cells'
// edges bulge by more than 0.05 index units at the latitude the boxes are used to, keep the pyramid search: it makes no such assumption.
bool mpg_grid_has_inverse(const mpg_grid_s *g, int stagger) {
  if (!g->has_inverse || stagger < 0 || stagger > 3 || !g->inverse_ok[stagger]) return false;
  const ProjDev &p = g->proj;
  if (p.code == MPG_PROJ_LC || p.code == MPG_PROJ_PS) return TG_EARTH_RADIUS_M / p.rebydx <= 200e3;
  if (p.code == MPG_PROJ_MERC)   // ... and a grid well short of the full circle (its cut sits opposite its middle column), true latitude <= 60
    return TG_EARTH_RADIUS_M / p.rebydx <= 200e3 && fabs(p.truelat1) <= 60.0 && (double)g->nx * p.dlon * TG_DEG_PER_RAD <= 300.0;
  if (p.code == MPG_PROJ_LATLON) {
    const double dlon = fabs(p.loninc), dlat = fabs(p.latinc);
    if (dlon > 2.0 || dlat > 2.0) return false;
    const double phi = MPG_LATLON_BOX_LIMIT * TG_RAD_PER_DEG;
    const double bulge = (atan(tan(phi) / cos(0.5 * dlon * TG_RAD_PER_DEG)) - phi) * TG_DEG_PER_RAD / dlat;
    return bulge <= 0.05;
  }
  return false;
}

// latlon_limit: the latitude (degrees) up to which a lat-lon grid's inverse is handed out -- MPG_LATLON_BOX_LIMIT (85) for the index
// BOXES of the Stores, whose pad follows the figure's latitude (geom.h mpg_box_pad); higher for the nearest search, which only
// places points
// unwrap_i: lat-lon indices are taken on the branch nearest to the grid's middle column instead of the reference's wrap rule
int mpg_k_points_ij(const mpg_grid_s *g, int64_t n, const double *x, const double *y, const double *z, float *ij, hipStream_t s, double latlon_limit,
                    bool unwrap_i) {
  const double i_center = (unwrap_i || g->proj.code == MPG_PROJ_MERC) ? 1.0 + 0.5 * (double)g->nx : (double)NAN;
  if (n > 0) k_points_ij<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(g->proj, g->proj_row0, latlon_limit, i_center, n, x, y, z, ij);
  MPG_HIP(hipGetLastError());
  return MPG_SUCCESS;
}

// does the inverse projection put the grid's own points of one stagger where they are?  A sample of up to 4096 of them (snx points
// per row); point (ii, jj) of the stagger belongs at CENTER index (ii - oi, jj - oj) -- oi = 0.5 for EDGE1 / CORNER, oj = 0.5 for EDGE2 /
// CORNER; bad += points that land more than 0.02 index units off (points in the zones where the inverse is not used are skipped, and
// so is the duplicate last column of a periodic grid's EDGE1 / CORNER stagger)
__global__ __launch_bounds__(256) void k_check_inverse(int snx, int64_t npts, int64_t step, float oi, float oj, int skip_col, const float *__restrict__ ij,
                                                       int32_t *__restrict__ bad) {
  const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, p = k * step;
  if (p >= npts) return;
  const float i = ij[2 * k], j = ij[2 * k + 1];
  if (i != i || j != j) return;
  if ((int)(p % snx) == skip_col) return;
  if (fabsf(i - ((float)(p % snx) - oi)) > 0.02f || fabsf(j - ((float)(p / snx) - oj)) > 0.02f) atomicAdd(bad, 1);
}

// Lower bound of the chord length (unit sphere) of ONE index unit of the grid, anywhere within `margin` index units of the
// latitudes lat_lo .. lat_hi (degrees) the grid's points span: the nearest-neighbour search in index space turns "every cell
// not looked at is at least r index units away" into "... at least r * h away on the sphere" with it (k_store_nearest.hip).
// Lambert: ground length per index unit = dx / m(lat), m the map factor (convex in latitude: its maximum over an interval
// sits at an end); lat-lon: min(dlat, dlon * cos(lat)) at the poleward end, capped at the 75 degrees beyond which the
// inverse is not used.  0: no bound (the caller keeps the BVH search).
double mpg_grid_min_index_chord(const mpg_grid_s *g, double lat_lo, double lat_hi, double margin) {
  const ProjDev &p = g->proj;
  if (p.code == MPG_PROJ_LC || p.code == MPG_PROJ_PS || p.code == MPG_PROJ_MERC) {
    // (polar stereographic: m falls monotonically towards the projection's pole; Mercator: m = cos(truelat) / cos(lat) grows with
    // |lat| -- either way the largest map factor of a latitude interval sits at one of its ends, as for Lambert.  An index unit is
    // dx / m on the ground along BOTH axes of these conformal maps.)
    const double mmin = p.code == MPG_PROJ_LC ? 1.0 : p.code == MPG_PROJ_PS ? 0.5 : cos(p.truelat1 * TG_RAD_PER_DEG);   // lower bound of m: the margin in degrees
    const double dlat = margin / (p.rebydx * fmax(mmin, 0.25)) * TG_DEG_PER_RAD * 1.5;   // index units -> degrees of latitude, generously
    const double a = fmax(lat_lo - dlat, -89.0), b = fmin(lat_hi + dlat, 89.0);
    const double m = fmax(map_factor(p, a), map_factor(p, b));
    if (!(m > 0.0) || !(m < 50.0)) return 0.0;
    return 1.0 / (p.rebydx * m);
  }
  if (p.code == MPG_PROJ_LATLON) {
    const double dlat = fabs(p.latinc), dlon = fabs(p.loninc);
    const double top = fmin(fmax(fabs(lat_lo - margin * dlat), fabs(lat_hi + margin * dlat)), 75.0);
    return fmin(dlat, dlon * cos(top * TG_RAD_PER_DEG)) * TG_RAD_PER_DEG;
  }
  return 0.0;
}

// The widest figure (index units) the Stores send through its index box: six degrees / 600 km across at most (the pads are
// small-angle bounds), and never more than 16 index units (beyond that the box holds hundreds of candidates and the walk wins)
double mpg_grid_box_emax(const mpg_grid_s *g) {
  const ProjDev &p = g->proj;
  double cell_deg = p.code != MPG_PROJ_LATLON ? (1.0 / p.rebydx) * TG_DEG_PER_RAD : fmax(fabs(p.loninc), fabs(p.latinc));
  // an index unit is dx / m on the ground; m falls to (1 + sin truelat) / 2 at a stereographic pole and to cos(truelat) on Mercator's equator
  if (p.code == MPG_PROJ_PS) cell_deg *= 2.0 / (1.0 + sin(fabs(p.truelat1) * TG_RAD_PER_DEG));
  if (p.code == MPG_PROJ_MERC) cell_deg /= cos(p.truelat1 * TG_RAD_PER_DEG);
  return fmin(16.0, 6.0 / cell_deg);
}

// How far the image of a figure can bulge out of the index-space box of its vertices, per squared index extent E^2 (the callers
// pad their boxes by 0.05 + coef * E^2 index units; 0.05 covers the float32 indices).  Lambert (conformal): a great-circle arc L
// grid lengths long bends by ~ L^2 * (grid length / earth radius) / 8 times a factor below 1 from the map scale's gradient --
// the coefficient is four times that.  Lat-lon (not conformal): the image of a great circle has coordinate curvature up to
// ~ 2 tan(lat) * (dlon/ds) * (dlat/ds); over an arc spanning E_i x E_j index units that is a deviation of ~ tan(lat) * E_i * E_j *
// delta / 4 radians -- it depends on the figure's latitude, so the kernels compute it per figure from mpg_grid_box_pad_latlon
// (geom.h mpg_box_pad) and this coefficient is not used for lat-lon grids.
double mpg_grid_box_pad_coef(const mpg_grid_s *g) {
  // conformal maps: the image of a great-circle arc of E index units deviates from its chord by (d ln m / d lat) / m * E^2 / (8 rebydx)
  // index units.  (d ln m / d lat) / m: Lambert below 1 where the inverse is handed out; polar stereographic cos(lat) / (1 + sin truelat)
  // <= 1; Mercator sin(lat) / cos(truelat) <= 1 / cos(truelat) -- hence its extra factor.  0.5 / rebydx is four times E^2 / (8 rebydx).
  if (g->proj.code == MPG_PROJ_LC || g->proj.code == MPG_PROJ_PS) return 0.5 / g->proj.rebydx;
  if (g->proj.code == MPG_PROJ_MERC) return 0.5 / (g->proj.rebydx * cos(g->proj.truelat1 * TG_RAD_PER_DEG));
  return 2.0 * fmax(fabs(g->proj.loninc), fabs(g->proj.latinc)) * TG_RAD_PER_DEG;
}
// radians per index unit of a lat-lon grid (0 for the other projections): the scale of the per-figure pad
double mpg_grid_box_pad_latlon(const mpg_grid_s *g) {
  if (g->proj.code != MPG_PROJ_LATLON) return 0.0;
  return fmax(fabs(g->proj.loninc), fabs(g->proj.latinc)) * TG_RAD_PER_DEG;
}

// stagger: MPG_STAGGERLOC_*; snx x sny points of that stagger
__global__ __launch_bounds__(256) void k_target_points(ProjDev p, int stagger, int snx, int sny, double *__restrict__ lon,
                                                       double *__restrict__ lat, double *__restrict__ x, double *__restrict__ y,
                                                       double *__restrict__ z, double *__restrict__ mapfac) {
  int64_t n = (int64_t)snx * sny;
  int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= n) return;
  double la, lo;
  {
#pragma clang fp contract(off)
    double fi = (double)(q % snx + 1), fj = (double)(q / snx + 1);
    double xi = (fi - 0.5) + 0.5, yj = (fj - 0.5) + 0.5;  // get_lat_lon_fields with sub_x = sub_y = 1
    if (stagger == MPG_STAGGERLOC_EDGE1 || stagger == MPG_STAGGERLOC_CORNER) xi -= 0.5;
    if (stagger == MPG_STAGGERLOC_EDGE2 || stagger == MPG_STAGGERLOC_CORNER) yj -= 0.5;
    ij_to_latlon(p, xi, yj, &la, &lo);
  }
  lon[q] = lo;
  lat[q] = la;
  // same conversion as k_grid_coords (k_setup.hip): the Stores see identical unit vectors for identical degrees
  const double d2r = 3.141592653589793 / 180.0;
  double sl, cl, so, co;
  sincos(la * d2r, &sl, &cl);
  sincos(lo * d2r, &so, &co);
  x[q] = cl * co;
  y[q] = cl * so;
  z[q] = sl;
  if (mapfac) mapfac[q] = map_factor(p, la);
}

// get_rotang: centred differences in j, one-sided on the first / last row
__global__ __launch_bounds__(256) void k_rotang(int nx, int ny, const double *__restrict__ lon, const double *__restrict__ lat,
                                                double *__restrict__ cosa, double *__restrict__ sina) {
#pragma clang fp contract(off)
  int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= (int64_t)nx * ny) return;
  int j = (int)(q / nx);
  int64_t jm = j > 0 ? q - nx : q, jp = j < ny - 1 ? q + nx : q;
  double d_lon = lon[jp] - lon[jm];
  if (d_lon > 180.0) d_lon -= 360.0;
  else if (d_lon < -180.0) d_lon += 360.0;
  double alpha = atan2(-cos(lat[q] * TG_RAD_PER_DEG) * (d_lon * TG_RAD_PER_DEG), (lat[jp] - lat[jm]) * TG_RAD_PER_DEG);
  sina[q] = sin(alpha);
  cosa[q] = cos(alpha);
}

static double wrap180(double x) {
  for (int it = 0; fabs(x) > 180.0 && it < 10; ++it) {
    if (x < -180.0) x += 360.0;
    if (x > 180.0) x -= 360.0;
  }
  return x;
}

// map_set / set_lc / lc_cone (module_map_utils.F90:1083-1157) on the host: a handful of scalars
static int derive(const mpg_proj *in, ProjDev *p) {
  memset(p, 0, sizeof(*p));
  p->code = in->code;
  p->lat1 = in->known_lat;
  p->lon1 = wrap180(in->known_lon);
  p->knowni = in->known_x;
  p->knownj = in->known_y;
  p->hemi = 1.0;
  if (in->code == MPG_PROJ_LC) {
    if (!(in->dx_m > 0.0) || fabs(in->truelat1) > 90.0) {
      mpg_set_error("mpg_grid_create_proj: Lambert needs dx_m > 0 and |truelat1| <= 90");
      return MPG_ERR_INVALID_ARG;
    }
    p->stdlon = wrap180(in->stand_lon);
    p->truelat1 = in->truelat1;
    p->truelat2 = fabs(in->truelat2) > 90.0 ? in->truelat1 : in->truelat2;
    p->hemi = in->truelat1 < 0.0 ? -1.0 : 1.0;
    p->rebydx = TG_EARTH_RADIUS_M / in->dx_m;
    if (fabs(p->truelat1 - p->truelat2) > 0.1) {
      double cone = log10(cos(p->truelat1 * TG_RAD_PER_DEG)) - log10(cos(p->truelat2 * TG_RAD_PER_DEG));
      cone = cone / (log10(tan((45.0 - fabs(p->truelat1) / 2.0) * TG_RAD_PER_DEG)) -
                     log10(tan((45.0 - fabs(p->truelat2) / 2.0) * TG_RAD_PER_DEG)));
      p->cone = cone;
    } else {
      p->cone = sin(fabs(p->truelat1) * TG_RAD_PER_DEG);
    }
    double deltalon1 = p->lon1 - p->stdlon;
    if (deltalon1 > 180.0) deltalon1 -= 360.0;
    if (deltalon1 < -180.0) deltalon1 += 360.0;
    double ctl1r = cos(p->truelat1 * TG_RAD_PER_DEG);
    double rsw = p->rebydx * ctl1r / p->cone *
                 pow(tan((90.0 * p->hemi - p->lat1) * TG_RAD_PER_DEG / 2.0) / tan((90.0 * p->hemi - p->truelat1) * TG_RAD_PER_DEG / 2.0),
                     p->cone);
    double arg = p->cone * (deltalon1 * TG_RAD_PER_DEG);
    p->polei = p->hemi * p->knowni - p->hemi * rsw * sin(arg);
    p->polej = p->hemi * p->knownj + rsw * cos(arg);
  } else if (in->code == MPG_PROJ_PS) {    // map_set + set_ps (module_map_utils.F90:682-715)
    if (!(in->dx_m > 0.0) || fabs(in->truelat1) > 90.0) {
      mpg_set_error("mpg_grid_create_proj: polar stereographic needs dx_m > 0 and |truelat1| <= 90");
      return MPG_ERR_INVALID_ARG;
    }
    p->stdlon = wrap180(in->stand_lon);
    p->truelat1 = in->truelat1;
    p->hemi = in->truelat1 < 0.0 ? -1.0 : 1.0;
    p->rebydx = TG_EARTH_RADIUS_M / in->dx_m;
    double reflon = p->stdlon + 90.0;
    double scale_top = 1.0 + p->hemi * sin(p->truelat1 * TG_RAD_PER_DEG);
    double ala1 = p->lat1 * TG_RAD_PER_DEG;
    double rsw = p->rebydx * cos(ala1) * scale_top / (1.0 + p->hemi * sin(ala1));
    double alo1 = (p->lon1 - reflon) * TG_RAD_PER_DEG;
    p->rsw = rsw;
    p->polei = p->knowni - rsw * cos(alo1);
    p->polej = p->knownj - p->hemi * rsw * sin(alo1);
  } else if (in->code == MPG_PROJ_MERC) {  // map_set + set_merc (:1293-1317)
    if (!(in->dx_m > 0.0) || fabs(in->truelat1) >= 90.0) {
      mpg_set_error("mpg_grid_create_proj: Mercator needs dx_m > 0 and |truelat1| < 90");
      return MPG_ERR_INVALID_ARG;
    }
    p->truelat1 = in->truelat1;
    p->hemi = in->truelat1 < 0.0 ? -1.0 : 1.0;
    p->rebydx = TG_EARTH_RADIUS_M / in->dx_m;
    double clain = cos(TG_RAD_PER_DEG * p->truelat1);
    p->dlon = in->dx_m / (TG_EARTH_RADIUS_M * clain);
    p->rsw = 0.0;
    if (p->lat1 != 0.0) p->rsw = log(tan(0.5 * ((p->lat1 + 90.0) * TG_RAD_PER_DEG))) / p->dlon;
  } else if (in->code == MPG_PROJ_LATLON) {
    if (in->dlat_deg == 0.0 || !(in->dlon_deg > 0.0)) {
      mpg_set_error("mpg_grid_create_proj: lat-lon needs dlat_deg != 0 and dlon_deg > 0");
      return MPG_ERR_INVALID_ARG;
    }
    p->latinc = in->dlat_deg;
    p->loninc = in->dlon_deg;
    p->nxmin = 1;
    p->nxmax = (int)nearbyint(360.0 / in->dlon_deg);
  } else {
    mpg_set_error("mpg_grid_create_proj: projection code %d not supported (PROJ_LATLON, PROJ_LC, PROJ_PS, PROJ_MERC)", in->code);
    return MPG_ERR_UNSUPPORTED;
  }
  return MPG_SUCCESS;
}

int mpg_k_target_grid(const mpg_proj *proj, mpg_grid_s *g, hipStream_t s) {
  ProjDev p;
  int rc = derive(proj, &p);
  if (rc) return rc;
  g->from_proj = true;
  g->proj_code = p.code;
  g->proj = p;
  g->has_inverse = true;
  g->proj_row0 = 0;
  for (int st = 0; st < 4; ++st) {
    int64_t n = (int64_t)g->snx[st] * g->sny[st];
    if ((rc = g->pts[st].alloc(n)) || (rc = g->lon[st].alloc(n)) || (rc = g->lat[st].alloc(n))) return rc;
    double *mf = nullptr;
    if (st != MPG_STAGGERLOC_CORNER) {
      if ((rc = g->mapfac[st].alloc(n))) return rc;
      mf = g->mapfac[st].p;
    }
    k_target_points<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(p, st, g->snx[st], g->sny[st], g->lon[st].p, g->lat[st].p,
                                                              g->pts[st].x.p, g->pts[st].y.p, g->pts[st].z.p, mf);
    MPG_HIP(hipGetLastError());
  }
  if (p.code == MPG_PROJ_LC) {  // model_grid.F90:1113
    int64_t n = (int64_t)g->nx * g->ny;
    if ((rc = g->cosa.alloc(n)) || (rc = g->sina.alloc(n))) return rc;
    k_rotang<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(g->nx, g->ny, g->lon[0].p, g->lat[0].p, g->cosa.p, g->sina.p);
    MPG_HIP(hipGetLastError());
  }
  MPG_HIP(hipStreamSynchronize(s));
  return MPG_SUCCESS;
}

// mpg_init loads this translation unit's code object ahead of its first launch (mpg_api.hip: warm_modules)
const void *mpg_anchor_k_target_grid() { return (const void *)k_target_points; }

// mpg_grid_attach_proj: a grid made from coordinate ARRAYS (mpg_grid_create) is told which projection those arrays came from
// -- rows row0 .. of that projection's grid, e.g. a rank's block of target rows -- so that the Stores can use the inverse
// projection as their candidate search.  The claim is CHECKED: a sample of the grid's own CENTER points must land on their own
// indices; a projection that does not fit is refused (a wrong one would make the Stores miss candidates).
__global__ __launch_bounds__(256) void k_sample_points(int64_t n, int64_t step, const double *__restrict__ x, const double *__restrict__ y,
                                                       const double *__restrict__ z, double *__restrict__ sx, double *__restrict__ sy,
                                                       double *__restrict__ sz) {
  const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (k * step >= n) return;
  sx[k] = x[k * step];
  sy[k] = y[k * step];
  sz[k] = z[k * step];
}
int mpg_k_attach_proj(mpg_grid_s *g, const mpg_proj *proj, int row0, hipStream_t s) {
  ProjDev p;
  int rc = derive(proj, &p);
  if (rc) return rc;
  const double i_center = p.code == MPG_PROJ_MERC ? 1.0 + 0.5 * (double)g->nx : (double)NAN;
  bool ok[4] = {true, true, true, true};
  int32_t nbad_center = 0;
  int64_t ns_center = 0;
  // every stagger that has points is checked: CENTER must fit (else the claim is refused); another stagger that does not -- the
  // CORNER points a file-defined grid gets from get_cell_corners sit a cell east of the projection's -- just keeps the pyramid
  // search for the Stores onto it
  for (int st = 0; st < 4; ++st) {
    const PointSet &c = g->pts[st];
    const int64_t n = (int64_t)g->snx[st] * g->sny[st];
    if (c.n != n || n == 0) continue;
    const int64_t step = n > 4096 ? n / 4096 : 1, ns = (n + step - 1) / step;
    TmpBuf<double> sp;
    TmpBuf<float> ij;
    TmpBuf<int32_t> bad;
    if ((rc = sp.alloc(3 * (size_t)ns, s)) || (rc = ij.alloc(2 * (size_t)ns, s)) || (rc = bad.alloc(1, s))) return rc;
    MPG_HIP(hipMemsetAsync(bad.p, 0, sizeof(int32_t), s));
    k_sample_points<<<(unsigned)((ns + 255) / 256), 256, 0, s>>>(n, step, c.x.p, c.y.p, c.z.p, sp.p, sp.p + ns, sp.p + 2 * ns);
    k_points_ij<<<(unsigned)((ns + 255) / 256), 256, 0, s>>>(p, row0, MPG_LATLON_BOX_LIMIT, i_center, ns, sp.p, sp.p + ns, sp.p + 2 * ns, ij.p);
    const float oi = (st == MPG_STAGGERLOC_EDGE1 || st == MPG_STAGGERLOC_CORNER) ? 0.5f : 0.f, oj = (st == MPG_STAGGERLOC_EDGE2 || st == MPG_STAGGERLOC_CORNER) ? 0.5f : 0.f;
    const int skip_col = ((g->periodic & MPG_GRID_PERIODIC_I) && oi > 0.f) ? g->nx : -1;
    k_check_inverse<<<(unsigned)((ns + 255) / 256), 256, 0, s>>>(g->snx[st], n, step, oi, oj, skip_col, ij.p, bad.p);
    MPG_HIP(hipGetLastError());
    int32_t hbad = 0;
    MPG_HIP(hipMemcpyAsync(&hbad, bad.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MPG_HIP(hipStreamSynchronize(s));
    ok[st] = hbad == 0;
    if (st == MPG_STAGGERLOC_CENTER) {
      nbad_center = hbad;
      ns_center = ns;
    }
  }
  if (!ok[MPG_STAGGERLOC_CENTER]) {
    mpg_set_error("mpg_grid_attach_proj: the projection does not reproduce the grid's own points (%d of %lld sampled CENTER points land elsewhere)",
                  nbad_center, (long long)ns_center);
    return MPG_ERR_INVALID_ARG;
  }
  g->proj = p;
  g->proj_row0 = row0;
  g->has_inverse = true;
  for (int st = 0; st < 4; ++st) g->inverse_ok[st] = ok[st];
  return MPG_SUCCESS;
}
