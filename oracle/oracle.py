"""ctypes front-end of the CPU oracle (oracle/mpassit_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package (mpassit_amd/).  Parity status: see the header of
mpassit_oracle.c ("parity unpinned" at the ESMF boundary; projection pinned by SURVEY App. E).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmpassit_oracle.so")
_lib = None

f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")

M, U, V, CORNER = 1, 2, 3, 4  # stagger codes of orc_xytoll


def build(force=False):
    src = os.path.join(_HERE, "mpassit_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_dual_triangles.restype = C.c_int64
        L.orc_conserve.restype = C.c_int64
        L.orc_fan_triangles.restype = C.c_int64
        L.orc_sizeof_proj.restype = C.c_int
        _lib = L
    return _lib


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def mesh_coords_deg(lon_rad, lat_rad):
    lon_rad, lat_rad = _c(lon_rad, np.float64), _c(lat_rad, np.float64)
    lo, la = np.empty_like(lon_rad), np.empty_like(lat_rad)
    lib().orc_mesh_coords_deg(C.c_int64(lon_rad.size), lon_rad.ctypes, lat_rad.ctypes, lo.ctypes, la.ctypes)
    return lo, la


def lonlat_deg_to_xyz(lon_deg, lat_deg):
    lon_deg, lat_deg = _c(lon_deg, np.float64).ravel(), _c(lat_deg, np.float64).ravel()
    xyz = np.empty((lon_deg.size, 3))
    lib().orc_lonlat_deg_to_xyz(C.c_int64(lon_deg.size), lon_deg.ctypes, lat_deg.ctypes, xyz.ctypes)
    return xyz


def dual_triangles(voc, n_vertices, cell_xyz):
    voc = _c(voc, np.int32)
    n_cells, max_edges = voc.shape
    tri = np.empty((n_vertices, 3), np.int32)
    cell_xyz = _c(cell_xyz, np.float64)
    n = lib().orc_dual_triangles(C.c_int64(n_cells), C.c_int64(n_vertices), C.c_int(max_edges), voc.ctypes,
                                 cell_xyz.ctypes, tri.ctypes)
    return tri, int(n)


def fan_triangles(voc, vert_xyz, origin=0):
    """Fan triangulation of the Voronoi polygons (node-located bilinear, App. A3): [nCells*(maxEdges-2)][3] vertex ids.
    origin: the fan's apex is listed vertex number origin mod n (-1: the last one) -- the kernels' "node_fan_origin" knob."""
    voc, vert_xyz = _c(voc, np.int32), _c(vert_xyz, np.float64)
    n_cells, max_edges = voc.shape
    ftri = np.empty((n_cells * (max_edges - 2), 3), np.int32)
    lib().orc_set_fan_origin(C.c_int(int(origin)))
    try:
        n = lib().orc_fan_triangles(C.c_int64(n_cells), C.c_int(max_edges), voc.ctypes, vert_xyz.ctypes, ftri.ctypes)
    finally:
        lib().orc_set_fan_origin(C.c_int(0))
    return ftri, int(n)


def bilinear_weights(cell_xyz, tri, pt_xyz, linetype=0):
    """linetype 0: the destination point meets the triangle's plane along the ray from the sphere's centre (the default of
    oracle and kernels); 1: along the plane's normal (ESMF_LINETYPE_CART read literally) -- DESIGN.md s2."""
    cell_xyz, tri, pt_xyz = _c(cell_xyz, np.float64), _c(tri, np.int32), _c(pt_xyz, np.float64)
    P = pt_xyz.shape[0]
    idx, w = np.empty((P, 3), np.int32), np.empty((P, 3))
    lib().orc_set_linetype(C.c_int(int(linetype)))
    try:
        lib().orc_bilinear_weights(C.c_int64(cell_xyz.shape[0]), cell_xyz.ctypes, C.c_int64(tri.shape[0]), tri.ctypes,
                                   C.c_int64(P), pt_xyz.ctypes, idx.ctypes, w.ctypes)
    finally:
        lib().orc_set_linetype(C.c_int(0))
    return idx, w


def nearest(cell_xyz, pt_xyz, brute=False):
    cell_xyz, pt_xyz = _c(cell_xyz, np.float64), _c(pt_xyz, np.float64)
    P = pt_xyz.shape[0]
    idx = np.empty(P, np.int32)
    fn = lib().orc_nearest_brute if brute else lib().orc_nearest
    fn(C.c_int64(cell_xyz.shape[0]), cell_xyz.ctypes, C.c_int64(P), pt_xyz.ctypes, idx.ctypes)
    return idx


def conserve(voc, vert_xyz, nx, ny, corner_xyz):
    """CSR (rowptr[nx*ny+1], col, val) of first-order conservative weights."""
    voc, vert_xyz, corner_xyz = _c(voc, np.int32), _c(vert_xyz, np.float64), _c(corner_xyz, np.float64)
    n_cells, max_edges = voc.shape
    assert corner_xyz.shape[0] == (nx + 1) * (ny + 1)
    rowptr = np.empty(nx * ny + 1, np.int64)
    args = (C.c_int64(n_cells), C.c_int64(vert_xyz.shape[0]), C.c_int(max_edges), voc.ctypes, vert_xyz.ctypes,
            C.c_int(nx), C.c_int(ny), corner_xyz.ctypes, rowptr.ctypes)
    nnz = lib().orc_conserve(*args, None, None, C.c_int64(0))
    col, val = np.empty(nnz, np.int32), np.empty(nnz)
    lib().orc_conserve(*args, col.ctypes, val.ctypes, C.c_int64(nnz))
    return rowptr, col, val


def grid_bilinear(nx, ny, centre_xyz, stagger, dst_xyz, tol=1e-10):
    """tol: how far outside a quad's parametric range a stagger point still counts as inside (the kernels' "grid_inside_tol_exp" knob)."""
    centre_xyz, dst_xyz = _c(centre_xyz, np.float64), _c(dst_xyz, np.float64)
    P = dst_xyz.shape[0]
    idx, w = np.empty((P, 4), np.int32), np.empty((P, 4))
    lib().orc_set_grid_tol(C.c_double(float(tol)))
    try:
        lib().orc_grid_bilinear(C.c_int(nx), C.c_int(ny), centre_xyz.ctypes, C.c_int(stagger), dst_xyz.ctypes,
                                idx.ctypes, w.ctypes)
    finally:
        lib().orc_set_grid_tol(C.c_double(1e-10))
    return idx, w


def grid_bilinear_periodic(nx, ny, flags, centre_xyz, stagger, dst_xyz):
    """Grid -> Grid bilinear on a 1PeriDim / monopole grid -> (idx [P][4], w [P][4], pole_src0 [2][nxd], pole_w [2][nxd])."""
    centre_xyz, dst_xyz = _c(centre_xyz, np.float64), _c(dst_xyz, np.float64)
    P = dst_xyz.shape[0]
    nxd = nx + 1 if stagger == 1 else nx
    idx, w = np.empty((P, 4), np.int32), np.empty((P, 4))
    pole_src0, pole_w = np.zeros((2, nxd), np.int32), np.zeros((2, nxd))
    lib().orc_grid_bilinear_p(C.c_int(nx), C.c_int(ny), C.c_int(flags), centre_xyz.ctypes, C.c_int(stagger), dst_xyz.ctypes,
                              idx.ctypes, w.ctypes, pole_src0.ctypes, pole_w.ctypes)
    return idx, w, pole_src0, pole_w


def apply_grid_periodic(nx, idx, w, pole_src0, pole_w, src, nlev):
    """A7 for a monopole handle: the 4-point sum plus w_pole * mean(CENTER row), src [nlev][nsrc] -> [nlev][P]."""
    out = apply_fixed(idx, w, src, nlev)
    src = np.asarray(src, np.float64).reshape(nlev, -1)
    nxd = pole_w.shape[1]
    P = idx.shape[0]
    for slot, p0 in ((0, 0), (1, P - nxd)):
        for i in range(nxd):
            if pole_w[slot, i] != 0.0:
                s0 = int(pole_src0[slot, i])
                out[:, p0 + i] += pole_w[slot, i] * src[:, s0:s0 + nx].mean(axis=1)
    return out


def apply_fixed(idx, w, src, nlev, lev_fast=False):
    idx, w, src = _c(idx, np.int32), _c(w, np.float64), _c(src, np.float64)
    P, nnz = idx.shape
    nsrc = src.size // nlev
    dst = np.empty((nlev, P))
    lib().orc_apply_fixed(C.c_int(nnz), C.c_int64(P), idx.ctypes, w.ctypes, C.c_int64(nsrc), C.c_int(nlev),
                          C.c_int(int(lev_fast)), src.ctypes, dst.ctypes)
    return dst


def apply_nearest(idx, src, nlev, lev_fast=False):
    idx, src = _c(idx, np.int32), _c(src, np.float64)
    P = idx.size
    nsrc = src.size // nlev
    dst = np.empty((nlev, P))
    lib().orc_apply_nearest(C.c_int64(P), idx.ctypes, C.c_int64(nsrc), C.c_int(nlev), C.c_int(int(lev_fast)),
                            src.ctypes, dst.ctypes)
    return dst


def apply_csr(rowptr, col, val, src, nlev, lev_fast=False):
    rowptr, col, val, src = _c(rowptr, np.int64), _c(col, np.int32), _c(val, np.float64), _c(src, np.float64)
    P = rowptr.size - 1
    nsrc = src.size // nlev
    dst = np.empty((nlev, P))
    lib().orc_apply_csr(C.c_int64(P), rowptr.ctypes, col.ctypes, val.ctypes, C.c_int64(nsrc), C.c_int(nlev),
                        C.c_int(int(lev_fast)), src.ctypes, dst.ctypes)
    return dst


def apply3_mt(idx, w, src, nlev, dst=None):
    """Threaded 3-point apply (cpu_baseline 'port'); src is cell-fastest [nlev][nsrc]."""
    P = idx.shape[0]
    nsrc = src.size // nlev
    if dst is None:
        dst = np.empty((nlev, P))
    lib().orc_apply3_mt(C.c_int64(P), idx.ctypes, w.ctypes, C.c_int64(nsrc), C.c_int(nlev), src.ctypes, dst.ctypes)
    return dst


def rotate_winds(cosa, sina, u, v):
    """In-place on copies; u, v are [nlev][npts]."""
    cosa, sina = _c(cosa, np.float64).ravel(), _c(sina, np.float64).ravel()
    u, v = np.array(u, np.float64, order="C"), np.array(v, np.float64, order="C")
    nlev = u.size // cosa.size
    lib().orc_rotate_winds(C.c_int64(cosa.size), C.c_int(nlev), cosa.ctypes, sina.ctypes, u.ctypes, v.ctypes)
    return u, v


class Proj:
    def __init__(self):
        self.buf = C.create_string_buffer(lib().orc_sizeof_proj())

    @classmethod
    def lambert(cls, truelat1, truelat2, stdlon, lat1, lon1, knowni, knownj, dx):
        p = cls()
        lib().orc_map_set_lc(p.buf, *(C.c_double(x) for x in (truelat1, truelat2, stdlon, lat1, lon1, knowni, knownj, dx)))
        return p

    @classmethod
    def latlon(cls, lat1, lon1, knowni, knownj, latinc, loninc):
        p = cls()
        lib().orc_map_set_latlon(p.buf, *(C.c_double(x) for x in (lat1, lon1, knowni, knownj, latinc, loninc)))
        return p

    @classmethod
    def polar(cls, truelat1, stdlon, lat1, lon1, knowni, knownj, dx):
        p = cls()
        lib().orc_map_set_ps(p.buf, *(C.c_double(x) for x in (truelat1, stdlon, lat1, lon1, knowni, knownj, dx)))
        return p

    @classmethod
    def mercator(cls, truelat1, lat1, lon1, knowni, knownj, dx):
        p = cls()
        lib().orc_map_set_merc(p.buf, *(C.c_double(x) for x in (truelat1, lat1, lon1, knowni, knownj, dx)))
        return p

    def map_factor(self, lat):
        """get_map_factor at latitudes `lat` (array) for this projection."""
        lib().orc_map_factor.restype = C.c_double
        return np.array([lib().orc_map_factor(self.buf, C.c_double(float(v))) for v in np.asarray(lat, np.float64).ravel()]).reshape(np.shape(lat))

    def fields(self):
        names = ["code_pad", "lat1", "lon1", "knowni", "knownj", "dx", "stdlon", "truelat1", "truelat2", "hemi", "cone",
                 "polei", "polej", "rsw", "rebydx", "latinc", "loninc"]
        vals = np.frombuffer(self.buf.raw[:8 * len(names)], dtype=np.float64)
        return dict(zip(names[1:], vals[1:]))

    def xytoll(self, x, y, stagger=M):
        la, lo = C.c_double(), C.c_double()
        lib().orc_xytoll(self.buf, C.c_double(x), C.c_double(y), C.c_int(stagger), C.byref(la), C.byref(lo))
        return la.value, lo.value

    def latlon_to_ij(self, lat, lon):
        i, j = C.c_double(), C.c_double()
        lib().orc_latlon_to_ij(self.buf, C.c_double(lat), C.c_double(lon), C.byref(i), C.byref(j))
        return i.value, j.value

    def lat_lon_fields(self, ni, nj, stagger):
        lat, lon = np.empty((nj, ni)), np.empty((nj, ni))
        lib().orc_lat_lon_fields(self.buf, C.c_int(ni), C.c_int(nj), C.c_int(stagger), lat.ctypes, lon.ctypes)
        return lat, lon


def get_rotang(xlat, xlon):
    xlat, xlon = _c(xlat, np.float64), _c(xlon, np.float64)
    nj, ni = xlat.shape
    cosa, sina = np.empty_like(xlat), np.empty_like(xlat)
    lib().orc_get_rotang(C.c_int(ni), C.c_int(nj), xlat.ctypes, xlon.ctypes, cosa.ctypes, sina.ctypes)
    return cosa, sina


def para_range(n1, n2, nprocs, irank):
    a, b = C.c_int(), C.c_int()
    lib().orc_para_range(C.c_int(n1), C.c_int(n2), C.c_int(nprocs), C.c_int(irank), C.byref(a), C.byref(b))
    return a.value, b.value


# ---- output post-ops of write_target_data (write_data.F90:1339-1475), numpy restatement -----------------------
def post_cast(x, scale=1.0, offset=0.0):
    """float64 -> NF90_FLOAT as nf90_put_var converts it; T - 300 (:1343) and PHB*9.81 (:1418) are one float64
    operation each before the conversion."""
    x = np.asarray(x, np.float64)
    if scale != 1.0:
        x = x * scale
    if offset != 0.0:
        x = x + offset
    return x.astype(np.float32)


def post_layer_mean(phb):
    """Z_C(k-1) = 0.5*(PHB(k) + PHB(k-1)), k = 2..nzp1 (write_data.F90:1406-1412)."""
    phb = np.asarray(phb, np.float64)
    return (0.5 * (phb[1:] + phb[:-1])).astype(np.float32)


def post_ptop(p_hyd):
    """write_data.F90:1362-1371: start from maxval(P_HYD); every column whose top value is >= 10 lowers it to
    min(0.8*top, current)."""
    p_hyd = np.asarray(p_hyd, np.float64)
    top = p_hyd[-1].ravel()
    cand = top[top >= 10.0] * 0.80
    v = p_hyd.max()
    if cand.size:
        v = min(v, cand.min())
    return np.float32(v)


def map_factor(lat_deg, proj_code, truelat1, truelat2):
    """get_map_factor (model_grid.F90:2229-2365) for PROJ_LC (= 1); PROJ_LATLON (= 0) has no branch in the reference
    (the arrays stay unset there), the product returns 1.0."""
    lat_deg = np.asarray(lat_deg, np.float64)
    if proj_code != 1:
        return np.ones_like(lat_deg)
    rad = 3.141592653589793 / 180.0
    colat = rad * (90.0 - lat_deg)
    if truelat1 != truelat2:
        c1, c2 = rad * (90.0 - truelat1), rad * (90.0 - truelat2)
        n = (np.log(np.sin(c1)) - np.log(np.sin(c2))) / (np.log(np.tan(c1 / 2.0)) - np.log(np.tan(c2 / 2.0)))
        return np.sin(c2) / np.sin(colat) * (np.tan(colat / 2.0) / np.tan(c2 / 2.0)) ** n
    c0 = rad * (90.0 - truelat1)
    return np.sin(c0) / np.sin(colat) * (np.tan(colat / 2.0) / np.tan(c0 / 2.0)) ** np.cos(c0)


def cell_corners(lat, lon, dx):
    """get_cell_corners (model_grid.F90:1902-1972), scalar loops in the reference's order of tests: far corner (bearing
    315 from (i_target, j_target)), extra column (225 from column i_target), extra row (45 from row j_target), all other
    points 135 from (i, j); d = sqrt(dx**2/2), pi = 3.14159265359, R = 6370000."""
    import math
    pi, R = 3.14159265359, 6370000.0
    nj, ni = lat.shape
    d = math.sqrt((dx ** 2.0) / 2.0)
    latc, lonc = np.empty((nj + 1, ni + 1)), np.empty((nj + 1, ni + 1))
    for j in range(nj + 1):
        for i in range(ni + 1):
            if j == nj and i == ni:
                sj, si, b = nj - 1, ni - 1, 315.0
            elif i == ni:
                sj, si, b = j, ni - 1, 225.0
            elif j == nj:
                sj, si, b = nj - 1, i, 45.0
            else:
                sj, si, b = j, i, 135.0
            lat1, lon1, brng = lat[sj, si] * (pi / 180.0), lon[sj, si] * (pi / 180.0), b * (pi / 180.0)
            lat2 = math.asin(math.sin(lat1) * math.cos(d / R) + math.cos(lat1) * math.sin(d / R) * math.cos(brng))
            lon2 = lon1 + math.atan2(math.sin(brng) * math.sin(d / R) * math.cos(lat1), math.cos(d / R) - math.sin(lat1) * math.sin(lat2))
            latc[j, i], lonc[j, i] = lat2 * 180.0 / pi, lon2 * 180.0 / pi
    return latc, lonc
