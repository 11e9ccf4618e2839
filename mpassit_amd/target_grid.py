"""Target-grid coordinates from namelist parameters (host side, numpy, float64).

Mirrors what the reference computes on the host before the hot path runs:
`define_target_grid_params` (model_grid.F90:644-1201) -> `get_lat_lon_fields` (:2188-2219) ->
`xytoll` (llxy_module.F90:166-216) -> `ij_to_latlon` (module_map_utils.F90:629-679; Lambert
:1160-1233 with `set_lc` :1083-1121 / `lc_cone` :1124-1157; lat-lon :1398-1428), plus `get_rotang`
(model_grid.F90:2450-2507) and the namelist-derived sizes of `read_setup_namelist`
(program_setup.F90:87-249: i_target = nx-1, j_target = ny-1, default known point = domain centre).

These coordinates are INPUTS of the HIP kernels ("generation is out of scope for HIP
but must be reproduced"); parity is pinned by the compiled-reference goldens of SURVEY App. E
(tests/golden/projection_lc.json).
"""
from dataclasses import dataclass, field

import numpy as np

PI = 3.141592653589793  # constants_module.F90:8
RAD_PER_DEG = PI / 180.0
DEG_PER_RAD = 180.0 / PI
EARTH_RADIUS_M = 6370000.0  # constants_module.F90:25
NAN = 1.0e20  # misc_definitions_module.F90:12 (namelist "unset" sentinel)

PROJ_LATLON, PROJ_LC, PROJ_PS, PROJ_MERC = 0, 1, 2, 3  # misc_definitions_module.F90:38-42
M, U, V, CORNER = 1, 2, 3, 6  # misc_definitions_module.F90:29


def _wrap180(x):
    it = 0
    while abs(x) > 180.0 and it < 10:
        if x < -180.0:
            x += 360.0
        if x > 180.0:
            x -= 360.0
        it += 1
    return x


@dataclass
class Proj:
    """`proj_info` subset (module_map_utils.F90:140-192) for PROJ_LC, PROJ_LATLON, PROJ_PS and PROJ_MERC."""
    code: int
    lat1: float = 0.0
    lon1: float = 0.0
    knowni: float = 0.0
    knownj: float = 0.0
    dx: float = 0.0
    stdlon: float = 0.0
    truelat1: float = 0.0
    truelat2: float = 0.0
    hemi: float = 1.0
    cone: float = 0.0
    polei: float = 0.0
    polej: float = 0.0
    rsw: float = 0.0
    rebydx: float = 0.0
    latinc: float = 0.0
    loninc: float = 0.0
    nxmin: int = 1
    nxmax: int = 0
    dlon: float = 0.0

    @staticmethod
    def lc_cone(truelat1, truelat2):
        if abs(truelat1 - truelat2) > 0.1:
            cone = np.log10(np.cos(truelat1 * RAD_PER_DEG)) - np.log10(np.cos(truelat2 * RAD_PER_DEG))
            cone = cone / (np.log10(np.tan((45.0 - abs(truelat1) / 2.0) * RAD_PER_DEG))
                           - np.log10(np.tan((45.0 - abs(truelat2) / 2.0) * RAD_PER_DEG)))
            return float(cone)
        return float(np.sin(abs(truelat1) * RAD_PER_DEG))

    @classmethod
    def lambert(cls, truelat1, truelat2, stdlon, lat1, lon1, knowni, knownj, dx):
        p = cls(PROJ_LC, lat1=lat1, lon1=_wrap180(lon1), knowni=knowni, knownj=knownj, dx=dx,
                stdlon=_wrap180(stdlon), truelat1=truelat1, truelat2=truelat2)
        p.hemi = -1.0 if truelat1 < 0.0 else 1.0
        p.rebydx = EARTH_RADIUS_M / dx
        if abs(p.truelat2) > 90.0:
            p.truelat2 = p.truelat1
        p.cone = cls.lc_cone(p.truelat1, p.truelat2)
        deltalon1 = p.lon1 - p.stdlon
        if deltalon1 > 180.0:
            deltalon1 -= 360.0
        if deltalon1 < -180.0:
            deltalon1 += 360.0
        ctl1r = np.cos(p.truelat1 * RAD_PER_DEG)
        p.rsw = float(p.rebydx * ctl1r / p.cone *
                      (np.tan((90.0 * p.hemi - p.lat1) * RAD_PER_DEG / 2.0) /
                       np.tan((90.0 * p.hemi - p.truelat1) * RAD_PER_DEG / 2.0)) ** p.cone)
        arg = p.cone * (deltalon1 * RAD_PER_DEG)
        p.polei = float(p.hemi * p.knowni - p.hemi * p.rsw * np.sin(arg))
        p.polej = float(p.hemi * p.knownj + p.rsw * np.cos(arg))
        return p

    @classmethod
    def polar(cls, truelat1, stdlon, lat1, lon1, knowni, knownj, dx):
        """map_set(PROJ_PS) + set_ps (module_map_utils.F90:682-715), arguments as push_source_projection passes them
        (llxy_module.F90:123-132)."""
        p = cls(PROJ_PS, lat1=lat1, lon1=_wrap180(lon1), knowni=knowni, knownj=knownj, dx=dx, stdlon=_wrap180(stdlon), truelat1=truelat1)
        p.hemi = -1.0 if truelat1 < 0.0 else 1.0
        p.rebydx = EARTH_RADIUS_M / dx
        reflon = p.stdlon + 90.0
        scale_top = 1.0 + p.hemi * np.sin(p.truelat1 * RAD_PER_DEG)
        ala1 = p.lat1 * RAD_PER_DEG
        p.rsw = float(p.rebydx * np.cos(ala1) * scale_top / (1.0 + p.hemi * np.sin(ala1)))
        alo1 = (p.lon1 - reflon) * RAD_PER_DEG
        p.polei = float(p.knowni - p.rsw * np.cos(alo1))
        p.polej = float(p.knownj - p.hemi * p.rsw * np.sin(alo1))
        return p

    @classmethod
    def mercator(cls, truelat1, lat1, lon1, knowni, knownj, dx):
        """map_set(PROJ_MERC) + set_merc (module_map_utils.F90:1293-1317; llxy_module.F90:71-79)."""
        p = cls(PROJ_MERC, lat1=lat1, lon1=_wrap180(lon1), knowni=knowni, knownj=knownj, dx=dx, truelat1=truelat1)
        p.hemi = -1.0 if truelat1 < 0.0 else 1.0
        p.rebydx = EARTH_RADIUS_M / dx
        clain = np.cos(RAD_PER_DEG * truelat1)
        p.dlon = float(dx / (EARTH_RADIUS_M * clain))
        p.rsw = 0.0
        if lat1 != 0.0:
            p.rsw = float(np.log(np.tan(0.5 * ((lat1 + 90.0) * RAD_PER_DEG))) / p.dlon)
        return p

    @classmethod
    def latlon(cls, lat1, lon1, knowni, knownj, latinc, loninc):
        return cls(PROJ_LATLON, lat1=lat1, lon1=_wrap180(lon1), knowni=knowni, knownj=knownj,
                   latinc=latinc, loninc=loninc, nxmin=1, nxmax=int(round(360.0 / loninc)))

    # -- ij_to_latlon, vectorised over numpy arrays of (i, j) in grid-index units (1-based)
    def ij_to_latlon(self, i, j):
        i = np.asarray(i, np.float64)
        j = np.asarray(j, np.float64)
        if self.code == PROJ_LC:
            chi1 = (90.0 - self.hemi * self.truelat1) * RAD_PER_DEG
            chi2 = (90.0 - self.hemi * self.truelat2) * RAD_PER_DEG
            xx = self.hemi * i - self.polei
            yy = self.polej - self.hemi * j
            r2 = xx * xx + yy * yy
            r = np.sqrt(r2) / self.rebydx
            lon = self.stdlon + DEG_PER_RAD * np.arctan2(self.hemi * xx, yy) / self.cone
            lon = np.fmod(lon + 360.0, 360.0)
            with np.errstate(divide="ignore", invalid="ignore"):
                if chi1 == chi2:
                    chi = 2.0 * np.arctan((r / np.tan(chi1)) ** (1.0 / self.cone) * np.tan(chi1 * 0.5))
                else:
                    chi = 2.0 * np.arctan((r * self.cone / np.sin(chi1)) ** (1.0 / self.cone) * np.tan(chi1 * 0.5))
            lat = (90.0 - chi * DEG_PER_RAD) * self.hemi
            pole = r2 == 0.0
            lat = np.where(pole, self.hemi * 90.0, lat)
            lon = np.where(pole, self.stdlon, lon)
            lon = np.where(lon > 180.0, lon - 360.0, lon)
            lon = np.where(lon < -180.0, lon + 360.0, lon)
            return lat, lon
        if self.code == PROJ_PS:                      # ijll_ps (module_map_utils.F90:763-822)
            reflon = self.stdlon + 90.0
            scale_top = 1.0 + self.hemi * np.sin(self.truelat1 * RAD_PER_DEG)
            xx = i - self.polei
            yy = (j - self.polej) * self.hemi
            r2 = xx * xx + yy * yy
            gi2 = (self.rebydx * scale_top) ** 2.0
            with np.errstate(divide="ignore", invalid="ignore"):
                lat = DEG_PER_RAD * self.hemi * np.arcsin((gi2 - r2) / (gi2 + r2))
                arccos = np.arccos(np.clip(xx / np.sqrt(r2), -1.0, 1.0))
            lon = np.where(yy > 0, reflon + DEG_PER_RAD * arccos, reflon - DEG_PER_RAD * arccos)
            pole = r2 == 0.0
            lat = np.where(pole, self.hemi * 90.0, lat)
            lon = np.where(pole, reflon, lon)
            lon = np.where(lon > 180.0, lon - 360.0, lon)
            lon = np.where(lon < -180.0, lon + 360.0, lon)
            return lat, lon
        if self.code == PROJ_MERC:                    # ijll_merc (:1344-1362)
            lat = 2.0 * np.arctan(np.exp(self.dlon * (self.rsw + j - self.knownj))) * DEG_PER_RAD - 90.0
            lon = (i - self.knowni) * self.dlon * DEG_PER_RAD + self.lon1
            lon = np.where(lon > 180.0, lon - 360.0, lon)
            lon = np.where(lon < -180.0, lon + 360.0, lon)
            return lat, lon
        span = float(self.nxmax - self.nxmin + 1)
        i_work = np.where(i < self.nxmin - 0.5, i + span, i)
        i_work = np.where(i >= self.nxmax + 0.5, i - span, i_work)
        lat = self.lat1 + (j - self.knownj) * self.latinc
        lon = self.lon1 + (i_work - self.knowni) * self.loninc
        return lat, lon

    def latlon_to_ij(self, lat, lon):
        """llij_lc (module_map_utils.F90:1236-1290), llij_ps (:718-760), llij_merc (:1320-1341); used by the synthetic
        meshes and the round-trip tests."""
        lat = np.asarray(lat, np.float64)
        lon = np.asarray(lon, np.float64)
        if self.code == PROJ_PS:
            reflon = self.stdlon + 90.0
            scale_top = 1.0 + self.hemi * np.sin(self.truelat1 * RAD_PER_DEG)
            ala = lat * RAD_PER_DEG
            rm = self.rebydx * np.cos(ala) * scale_top / (1.0 + self.hemi * np.sin(ala))
            alo = (lon - reflon) * RAD_PER_DEG
            return self.polei + rm * np.cos(alo), self.polej + self.hemi * rm * np.sin(alo)
        if self.code == PROJ_MERC:
            deltalon = lon - self.lon1
            deltalon = np.where(deltalon < -180.0, deltalon + 360.0, deltalon)
            deltalon = np.where(deltalon > 180.0, deltalon - 360.0, deltalon)
            i = self.knowni + (deltalon / (self.dlon * DEG_PER_RAD))
            j = self.knownj + np.log(np.tan(0.5 * ((lat + 90.0) * RAD_PER_DEG))) / self.dlon - self.rsw
            return i, j
        assert self.code == PROJ_LC
        deltalon = lon - self.stdlon
        deltalon = np.where(deltalon > 180.0, deltalon - 360.0, deltalon)
        deltalon = np.where(deltalon < -180.0, deltalon + 360.0, deltalon)
        ctl1r = np.cos(self.truelat1 * RAD_PER_DEG)
        rm = self.rebydx * ctl1r / self.cone * (np.tan((90.0 * self.hemi - lat) * RAD_PER_DEG / 2.0) /
                                                np.tan((90.0 * self.hemi - self.truelat1) * RAD_PER_DEG / 2.0)) ** self.cone
        arg = self.cone * (deltalon * RAD_PER_DEG)
        i = self.hemi * (self.polei + self.hemi * rm * np.sin(arg))
        j = self.hemi * (self.polej - rm * np.cos(arg))
        return i, j

    def xytoll(self, x, y, stagger=M):
        x = np.asarray(x, np.float64)
        y = np.asarray(y, np.float64)
        if stagger == U:
            x = x - 0.5
        elif stagger == V:
            y = y - 0.5
        elif stagger == CORNER:
            x, y = x - 0.5, y - 0.5
        return self.ij_to_latlon(x, y)

    def lat_lon_fields(self, ni, nj, stagger):
        """get_lat_lon_fields: arrays [nj][ni] (i fastest) for 1-based points (i, j)."""
        jj, ii = np.meshgrid(np.arange(1, nj + 1, dtype=np.float64), np.arange(1, ni + 1, dtype=np.float64), indexing="ij")
        return self.xytoll((ii - 0.5) + 0.5, (jj - 0.5) + 0.5, stagger)


def get_map_factor(proj, xlat):
    """MAPFAC at the points with latitude xlat (get_map_factor, model_grid.F90:2229-2365): Lambert (one or two true
    latitudes), polar stereographic, Mercator; lat-lon has no branch there (1.0 here, like the device kernel)."""
    xlat = np.asarray(xlat, np.float64)
    if proj.code == PROJ_LC:
        colat = RAD_PER_DEG * (90.0 - xlat)
        if proj.truelat1 != proj.truelat2:
            colat1, colat2 = RAD_PER_DEG * (90.0 - proj.truelat1), RAD_PER_DEG * (90.0 - proj.truelat2)
            n = (np.log(np.sin(colat1)) - np.log(np.sin(colat2))) / (np.log(np.tan(colat1 / 2.0)) - np.log(np.tan(colat2 / 2.0)))
            return np.sin(colat2) / np.sin(colat) * (np.tan(colat / 2.0) / np.tan(colat2 / 2.0)) ** n
        colat0 = RAD_PER_DEG * (90.0 - proj.truelat1)
        return np.sin(colat0) / np.sin(colat) * (np.tan(colat / 2.0) / np.tan(colat0 / 2.0)) ** np.cos(colat0)
    if proj.code == PROJ_PS:
        return (1.0 + np.sin(RAD_PER_DEG * abs(proj.truelat1))) / (1.0 + np.sin(RAD_PER_DEG * np.copysign(1.0, proj.truelat1) * xlat))
    if proj.code == PROJ_MERC:
        return np.sin(RAD_PER_DEG * (90.0 - proj.truelat1)) / np.sin(RAD_PER_DEG * (90.0 - xlat))
    return np.ones_like(xlat)


def get_rotang(xlat, xlon):
    """cos/sin of the grid rotation angle, arrays [nj][ni] (model_grid.F90:2450-2507)."""
    nj = xlat.shape[0]
    jm = np.maximum(np.arange(nj) - 1, 0)
    jp = np.minimum(np.arange(nj) + 1, nj - 1)
    d_lon = xlon[jp, :] - xlon[jm, :]
    d_lon = np.where(d_lon > 180.0, d_lon - 360.0, np.where(d_lon < -180.0, d_lon + 360.0, d_lon))
    alpha = np.arctan2(-np.cos(xlat * RAD_PER_DEG) * (d_lon * RAD_PER_DEG), (xlat[jp, :] - xlat[jm, :]) * RAD_PER_DEG)
    return np.cos(alpha), np.sin(alpha)


@dataclass
class TargetGrid:
    """Host copy of what `define_target_grid_params` leaves in the ESMF Grid (all [nj][ni], degrees)."""
    nx: int  # mass points west-east  (= i_target = namelist nx - 1)
    ny: int  # mass points south-north (= j_target)
    proj: Proj
    is_regional: bool
    lat: np.ndarray = None
    lon: np.ndarray = None
    lat_u: np.ndarray = None
    lon_u: np.ndarray = None
    lat_v: np.ndarray = None
    lon_v: np.ndarray = None
    lat_c: np.ndarray = None
    lon_c: np.ndarray = None
    cosa: np.ndarray = None
    sina: np.ndarray = None
    extra: dict = field(default_factory=dict)


def define_target_grid_params(target_grid_type, nx, ny, dx=NAN, dy=NAN, ref_lat=NAN, ref_lon=NAN, ref_x=NAN, ref_y=NAN,
                              truelat1=NAN, truelat2=NAN, stand_lon=NAN, is_regional=True, arrays=True):
    """Namelist (&config, program_setup.F90:103-106) -> TargetGrid.  nx, ny are the namelist's
    STAGGERED counts: the mass grid is (nx-1) x (ny-1) (program_setup.F90:163-164).
    arrays=False stops after the namelist checks and the projection set-up: the coordinate arrays are then
    produced on the device by `regrid.Grid.from_proj(target)`."""
    i_target, j_target = nx - 1, ny - 1
    kind = target_grid_type.upper()
    known_x, known_y, known_lat, known_lon = ref_x, ref_y, ref_lat, ref_lon
    if kind == "LAMBERT":
        if truelat2 == NAN:
            if truelat1 == NAN:
                raise ValueError("No TRUELAT1 specified for Lambert conformal projection.")
            truelat2 = truelat1
    elif kind == "LAT-LON":
        if dx == NAN and dy == NAN:
            if is_regional:
                raise ValueError("For lat-lon projection, if dx/dy are not specified a global grid is assumed.")
            dlondeg, dlatdeg = 360.0 / i_target, 180.0 / j_target
            known_x = known_y = 1.0
            known_lon = stand_lon + dlondeg / 2.0
            known_lat = -90.0 + dlatdeg / 2.0
        else:
            if not is_regional:
                raise ValueError("For lat-lon projection, if dx/dy are specified a regional grid is assumed.")
            dlatdeg, dlondeg = dy, dx
            if known_lat == NAN or known_lon == NAN:
                raise ValueError("For lat-lon projection with dx/dy, ref_lat/ref_lon must be specified")
    elif kind in ("MERCATOR", "POLAR"):
        if truelat1 == NAN:
            raise ValueError("No TRUELAT1 specified for the %s projection." % kind.lower())
    else:
        raise ValueError('In namelist, invalid target_grid_type specified. Valid projections are "lambert", "mercator", "polar", and '
                         '"lat-lon".')
    if known_x == NAN and known_y == NAN:
        known_x, known_y = (i_target + 1) / 2.0, (j_target + 1) / 2.0
    elif known_x == NAN or known_y == NAN:
        raise ValueError("In namelist, neither or both of ref_x, ref_y must be specified.")
    if kind == "LAMBERT":
        proj = Proj.lambert(truelat1, truelat2, stand_lon, known_lat, known_lon, known_x, known_y, dx)
    elif kind == "POLAR":
        proj = Proj.polar(truelat1, stand_lon, known_lat, known_lon, known_x, known_y, dx)
    elif kind == "MERCATOR":
        proj = Proj.mercator(truelat1, known_lat, known_lon, known_x, known_y, dx)
    else:
        proj = Proj.latlon(known_lat, known_lon, known_x, known_y, dlatdeg, dlondeg)
    g = TargetGrid(i_target, j_target, proj, is_regional)
    if not arrays:      # coordinates are generated on the device (regrid.Grid.from_proj -> mpg_grid_create_proj)
        return g
    g.lat, g.lon = proj.lat_lon_fields(i_target, j_target, M)
    g.lat_v, g.lon_v = proj.lat_lon_fields(i_target, j_target + 1, V)
    g.lat_u, g.lon_u = proj.lat_lon_fields(i_target + 1, j_target, U)
    g.lat_c, g.lon_c = proj.lat_lon_fields(i_target + 1, j_target + 1, CORNER)
    if proj.code == PROJ_LC:
        g.cosa, g.sina = get_rotang(g.lat, g.lon)
    return g


def get_cell_corners(lat, lon, dx):
    """CORNER-stagger coordinates of a file-defined target grid, exactly as get_cell_corners computes them
    (model_grid.F90:1902-1972): every corner is the great-circle destination point at distance sqrt(dx^2/2) from a
    mass point, with the bearings AS WRITTEN there -- 135 degrees for the (i, j) block, 225 on the extra column
    (from column i_target), 45 on the extra row (from row j_target), 315 at the far corner -- and its own constants
    pi = 3.14159265359, R = 6370000.  lat, lon: [nj][ni] degrees -> ([nj+1][ni+1], [nj+1][ni+1])."""
    pi, R = 3.14159265359, 6370000.0
    nj, ni = lat.shape
    d = np.sqrt((dx ** 2.0) / 2.0)

    def dest(la, lo, bearing_deg):
        lat1, lon1 = la * (pi / 180.0), lo * (pi / 180.0)
        brng = bearing_deg * (pi / 180.0)
        lat2 = np.arcsin(np.sin(lat1) * np.cos(d / R) + np.cos(lat1) * np.sin(d / R) * np.cos(brng))
        lon2 = lon1 + np.arctan2(np.sin(brng) * np.sin(d / R) * np.cos(lat1), np.cos(d / R) - np.sin(lat1) * np.sin(lat2))
        return lat2 * 180.0 / pi, lon2 * 180.0 / pi
    latc, lonc = np.empty((nj + 1, ni + 1)), np.empty((nj + 1, ni + 1))
    latc[:nj, :ni], lonc[:nj, :ni] = dest(lat, lon, 135.0)
    latc[:nj, ni], lonc[:nj, ni] = dest(lat[:, ni - 1], lon[:, ni - 1], 225.0)
    latc[nj, :ni], lonc[nj, :ni] = dest(lat[nj - 1, :], lon[nj - 1, :], 45.0)
    latc[nj, ni], lonc[nj, ni] = dest(lat[nj - 1, ni - 1], lon[nj - 1, ni - 1], 315.0)
    return latc, lonc


def define_target_grid_file(path):
    """target_grid_type = 'file' (define_target_grid_file, model_grid.F90:1203-1888): the grid comes from a WRF
    geo_em / wrfinput style file -- dimensions west_east / south_north, global attributes DX, CEN_LAT, CEN_LON,
    TRUELAT1/2, MOAD_CEN_LAT, STAND_LON, POLE_LAT/LON, MAP_PROJ, variables XLONG|XLONG_M, XLAT|XLAT_M, XLONG_U, XLAT_U,
    XLONG_V, XLAT_V, MAPFAC_M/U/V, SINALPHA / COSALPHA (Lambert), HGT|HGT_M; corners from get_cell_corners.
    Classic-format files only (ncio).  -> TargetGrid with host arrays (use regrid.Grid.from_target)."""
    from . import ncio
    with ncio.Reader(path) as r:
        ni, nj = r.dims["west_east"], r.dims["south_north"]

        def var(*names):
            for n in names:
                if n in r.vars:
                    v = r.vars[n]
                    return r.get(n, rec=0, dtype=np.float64) if v["record"] else r.get(n, dtype=np.float64)
            raise KeyError("%s: none of %s found" % (path, "/".join(names)))

        def att(name, default=NAN):
            try:
                return float(r.att(name)[0])
            except ncio.NcioError:
                return default
        dx = att("DX")
        code = int(att("MAP_PROJ", PROJ_LC))
        proj = Proj(code, dx=dx, stdlon=att("STAND_LON"), truelat1=att("TRUELAT1"), truelat2=att("TRUELAT2"))
        g = TargetGrid(ni, nj, proj, True)
        g.lon, g.lat = var("XLONG", "XLONG_M"), var("XLAT", "XLAT_M")
        g.lon_u, g.lat_u = var("XLONG_U"), var("XLAT_U")
        g.lon_v, g.lat_v = var("XLONG_V"), var("XLAT_V")
        g.lat_c, g.lon_c = get_cell_corners(g.lat, g.lon, dx)
        # The projection as a CLAIM the library checks on the grid's own points (mpg_grid_attach_proj: the Stores may then search
        # through its inverse instead of the box pyramid): the file's MAP_PROJ / TRUELAT1/2 / STAND_LON / DX with the grid's own first
        # mass point as the known point.  WRF numbers its projections 1 Lambert, 2 polar stereographic, 3 Mercator, 6 lat-lon
        # (a rotated lat-lon grid fails the check and keeps the pyramid).
        proj.lat1, proj.lon1, proj.knowni, proj.knownj = float(g.lat[0, 0]), float(g.lon[0, 0]), 1.0, 1.0
        if code == 6:
            proj.code = PROJ_LATLON
            proj.latinc = float(g.lat[1, 0] - g.lat[0, 0]) if nj > 1 else 0.0
            dlon = float(g.lon[0, 1] - g.lon[0, 0]) if ni > 1 else 0.0
            proj.loninc = dlon + 360.0 if dlon < -180.0 else dlon
        for k, names in (("mapfac_m", ("MAPFAC_M",)), ("mapfac_u", ("MAPFAC_U",)), ("mapfac_v", ("MAPFAC_V",)), ("hgt", ("HGT", "HGT_M"))):
            try:
                g.extra[k] = var(*names)
            except KeyError:
                pass
        if code == PROJ_LC:
            g.sina, g.cosa = var("SINALPHA"), var("COSALPHA")
        g.extra.update(from_file=True, ref_lat=att("MOAD_CEN_LAT", att("CEN_LAT")), ref_lon=att("CEN_LON"), pole_lat=att("POLE_LAT", 90.0),
                       pole_lon=att("POLE_LON", 0.0))
        try:                                                              # model_grid.F90:1288-1295
            mpc = r.att("MAP_PROJ_CHAR")
            g.extra["map_proj_char"] = mpc.strip() if isinstance(mpc, str) else None
        except ncio.NcioError:
            g.extra["map_proj_char"] = None
    return g
