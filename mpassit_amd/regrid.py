"""Host-side mirror of the ESMF verbs MPASSIT's hot path uses, bound to the HIP C-ABI.

Names follow the reference's call sites so that tests read like the reference:
  Mesh            <- ESMF_MeshCreate                 (model_grid.F90:488-497)
  Grid            <- ESMF_GridCreate* + GridAddCoord (model_grid.F90:684-728,736-1038)
  regrid_store    <- ESMF_Field[Bundle]RegridStore   (interp.F90:123,207,...,437)
  RouteHandle.regrid      <- ESMF_Field[Bundle]Regrid (interp.F90:134,219,...,443)
  RouteHandle.release     <- ESMF_FieldBundleRegridRelease (interp.F90:450-463)
Host numpy arrays go through mpg_regrid (H2D + kernel + D2H); torch CUDA tensors go through
mpg_regrid_dev on torch's current stream (device-resident fast path used by bench.py).
"""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import (LAYOUT_CELL_FAST, LAYOUT_LEV_FAST, MESHLOC_ELEMENT, MESHLOC_NODE, REGRIDMETHOD_BILINEAR, REGRIDMETHOD_CONSERVE,
                   REGRIDMETHOD_NEAREST_STOD, STAGGERLOC_CENTER, STAGGERLOC_CORNER, STAGGERLOC_EDGE1, STAGGERLOC_EDGE2,
                   check)

__all__ = ["MESHLOC_ELEMENT", "MESHLOC_NODE", "Mesh", "Grid", "RouteHandle", "regrid_store", "regrid_store_grid", "regrid_store_begin", "regrid_store_grid_begin", "rotate_winds_cgrid", "wind_destagger",
           "REGRIDMETHOD_BILINEAR", "REGRIDMETHOD_CONSERVE", "REGRIDMETHOD_NEAREST_STOD", "STAGGERLOC_CENTER",
           "STAGGERLOC_EDGE1", "STAGGERLOC_EDGE2", "STAGGERLOC_CORNER", "LAYOUT_CELL_FAST", "LAYOUT_LEV_FAST"]


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Algorithmic-byte accounting of a sequence of device calls (bench.py's `job` leg): while ACCOUNT is a list, every
# device-side Regrid / rotation appends (what, bytes) with the bytes of SURVEY s8(d) -- U*L*e_src + P*L*e_dst per field plus
# the handle's indices and weights once per call (U = the sources the handle references).  Off (None) by default: looking
# up U synchronises.
ACCOUNT = None


def _account_regrid(rh, nlev, nfields, src_elem, dst_elem):
    if ACCOUNT is None:
        return
    if getattr(rh, "_acc_U", None) is None:
        try:
            rh._acc_U = int(rh.unique_sources().size)
        except L.MpgError:                      # handles with pole terms: every source of the grid
            rh._acc_U = int(rh.n_src)
    wbytes = (rh.nnz * 12 + (rh.n_dst + 1) * 4) if rh.nnz_per_row == 0 else rh.n_dst * (4 if rh.nnz_per_row == 1 else 12 * rh.nnz_per_row)
    ACCOUNT.append(("regrid nnz%d L%d x%d" % (rh.nnz_per_row, nlev, nfields), nfields * nlev * (rh._acc_U * src_elem + rh.n_dst * dst_elem) + wbytes))


class Mesh:
    """MPAS mesh as the reference hands it to ESMF: elements = cells, nodes = vertices.
    lat/lon in radians (file convention), verticesOnCell [nCells][maxEdges] 1-based, 0-padded."""

    def __init__(self, latCell, lonCell, latVertex, lonVertex, verticesOnCell, window_grid=None):
        """window_grid: a Grid (one rank's row block) -- only the part of the mesh that grid can see is brought to the
        device (mpg_mesh_create_window); ids stay global, Stores onto that grid give the weights of the whole mesh."""
        latCell, lonCell, latVertex, lonVertex = map(_f64, (latCell, lonCell, latVertex, lonVertex))
        voc = np.ascontiguousarray(verticesOnCell, dtype=np.int32)
        if voc.ndim != 2 or voc.shape[0] != latCell.size:
            raise ValueError("verticesOnCell must be [nCells][maxEdges]")
        self.nCells, self.nVertices, self.maxEdges = int(latCell.size), int(latVertex.size), int(voc.shape[1])
        self._h = C.c_void_p()
        args = (C.c_int64(self.nCells), C.c_int64(self.nVertices), C.c_int(self.maxEdges), _ptr(latCell), _ptr(lonCell), _ptr(latVertex),
                _ptr(lonVertex), _ptr(voc))
        if window_grid is None:
            check(L.load().mpg_mesh_create(*args, C.byref(self._h)))
        else:
            check(L.load().mpg_mesh_create_window(*args, window_grid._h, C.byref(self._h)))

    @classmethod
    def from_mpas(cls, m, window_grid=None):
        return cls(m.latCell, m.lonCell, m.latVertex, m.lonVertex, m.verticesOnCell, window_grid=window_grid)

    def window_info(self):
        """(cell_first, cell_count, vertex_first, vertex_count, margin): the resident part of the geometry (the whole mesh
        unless it was made with window_grid) and the chord distance from the grid within which every cell is present."""
        a, b, c, d, mg = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_double()
        check(L.load().mpg_mesh_window_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(mg)))
        return a.value, b.value, c.value, d.value, mg.value

    def set_source_window(self, first, count, meshloc=MESHLOC_ELEMENT):
        """Every handle of this mesh and location (existing and future) indexes its sources relative to `first`; Regrid
        then reads slabs of `count` ids (mpg_mesh_set_source_window).  Route handle objects made earlier: call
        RouteHandle._refresh() (n_src changes)."""
        check(L.load().mpg_mesh_set_source_window(self._h, C.c_int(meshloc), C.c_int64(first), C.c_int64(count)))

    def triangles(self):
        tri = np.empty((self.nVertices, 3), np.int32)
        check(L.load().mpg_mesh_get_triangles(self._h, _ptr(tri)))
        return tri

    def destroy(self):
        if self._h:
            check(L.load().mpg_mesh_destroy(self._h))
            self._h = C.c_void_p()


class _Proj(C.Structure):
    """struct mpg_proj (include/mpassit_amd.h)."""
    _fields_ = [("code", C.c_int), ("known_lat", C.c_double), ("known_lon", C.c_double), ("known_x", C.c_double),
                ("known_y", C.c_double), ("dx_m", C.c_double), ("stand_lon", C.c_double), ("truelat1", C.c_double),
                ("truelat2", C.c_double), ("dlat_deg", C.c_double), ("dlon_deg", C.c_double)]


class Grid:
    """Structured target grid with its four staggers (degrees, arrays [nj][ni], i fastest)."""

    def __init__(self, lon, lat, lon_corner=None, lat_corner=None, lon_u=None, lat_u=None, lon_v=None, lat_v=None,
                 periodic=False):
        lon, lat = _f64(lon), _f64(lat)
        self.ny, self.nx = lat.shape
        arrs = [None if a is None else _f64(a) for a in (lon_corner, lat_corner, lon_u, lat_u, lon_v, lat_v)]
        shapes = [(self.ny + 1, self.nx + 1)] * 2 + [(self.ny, self.nx + 1)] * 2 + [(self.ny + 1, self.nx)] * 2
        for a, s in zip(arrs, shapes):
            if a is not None and a.shape != s:
                raise ValueError("stagger coordinate array has shape %s, expected %s" % (a.shape, s))
        self._h = C.c_void_p()
        check(L.load().mpg_grid_create(C.c_int(self.nx), C.c_int(self.ny), C.c_int(int(periodic)), _ptr(lon), _ptr(lat),
                                       *[_ptr(a) for a in arrs], C.byref(self._h)))

    @classmethod
    def from_target(cls, g, rows=None, attach_proj=True):
        """From target_grid.TargetGrid; rows=(j0, j1) keeps only mass rows [j0, j1) (multi-GPU row shard,
        mirrors the reference's regDecomp=(/1,npets/) split along j, model_grid.F90:693).  A global grid
        (is_regional=.false.) is periodic in i with monopole caps (model_grid.F90:685-694); a row block keeps only
        the caps it touches."""
        flags = 0 if g.is_regional else L.GRID_PERIODIC_I
        if rows is None:
            self = cls(g.lon, g.lat, g.lon_c, g.lat_c, g.lon_u, g.lat_u, g.lon_v, g.lat_v, periodic=flags)
            j0 = 0
        else:
            j0, j1 = rows
            if flags:
                flags |= (L.GRID_NO_SOUTH_POLE if j0 > 0 else 0) | (L.GRID_NO_NORTH_POLE if j1 < g.ny else 0)
            self = cls(g.lon[j0:j1], g.lat[j0:j1], g.lon_c[j0:j1 + 1], g.lat_c[j0:j1 + 1], g.lon_u[j0:j1], g.lat_u[j0:j1],
                       g.lon_v[j0:j1 + 1], g.lat_v[j0:j1 + 1], periodic=flags)
        if attach_proj and getattr(g, "proj", None) is not None:
            try:                               # the arrays came from this projection: the Stores may search through its inverse.
                self.attach_proj(g.proj, j0)   # The library checks the claim on the grid's own points; a projection that does not
            except L.MpgError:                 # reproduce them (a grid read from a file, whose Proj is incomplete) is refused and
                pass                           # the grid keeps the pyramid search
        return self

    def attach_proj(self, p, row0=0):
        """mpg_grid_attach_proj: the coordinate arrays of this grid are rows row0 .. of projection `p` (target_grid.Proj)."""
        c = _Proj(code=p.code, known_lat=p.lat1, known_lon=p.lon1, known_x=p.knowni, known_y=p.knownj, dx_m=p.dx,
                  stand_lon=p.stdlon, truelat1=p.truelat1, truelat2=p.truelat2, dlat_deg=p.latinc, dlon_deg=p.loninc)
        check(L.load().mpg_grid_attach_proj(self._h, C.byref(c), C.c_int(int(row0))))

    @classmethod
    def from_proj(cls, g, fill_target=True):
        """Device-side target grid (mpg_grid_create_proj): all four staggers, map factors and cos/sin(alpha) are
        computed on the GPU from the projection of target_grid.TargetGrid `g` (built with arrays=False or not).
        fill_target: copy cos/sin(alpha) back into `g` (interp's wind rotation reads them from there)."""
        p = g.proj
        c = _Proj(code=p.code, known_lat=p.lat1, known_lon=p.lon1, known_x=p.knowni, known_y=p.knownj, dx_m=p.dx,
                  stand_lon=p.stdlon, truelat1=p.truelat1, truelat2=p.truelat2, dlat_deg=p.latinc, dlon_deg=p.loninc)
        self = cls.__new__(cls)
        self.nx, self.ny = g.nx, g.ny
        self._h = C.c_void_p()
        check(L.load().mpg_grid_create_proj(C.byref(c), C.c_int(g.nx), C.c_int(g.ny),
                                            C.c_int(0 if g.is_regional else L.GRID_PERIODIC_I), C.byref(self._h)))
        self.built_from_proj = True
        if fill_target and p.code == 1:
            g.cosa, g.sina = self.rotang()
        return self

    def coords(self, staggerloc):
        """(lon, lat) in degrees of a projection-built grid, [nj][ni] of that stagger (XLONG/XLAT[_U,_V] of the file)."""
        lon, lat = np.empty(self.stagger_shape(staggerloc)), np.empty(self.stagger_shape(staggerloc))
        check(L.load().mpg_grid_get_coords(self._h, C.c_int(staggerloc), _ptr(lon), _ptr(lat)))
        return lon, lat

    def rotang_dev(self):
        """(cosalpha, sinalpha) as they sit in device memory, owned by the grid (mpg_grid_rotang_dev): DevArray views for rotate_winds_cgrid /
        wind_destagger -- no download and upload of 2 x ny x nx doubles (configuration 4: 30 MB through pageable memory, 1.5 ms of a cold job)."""
        c, s = C.c_void_p(), C.c_void_p()
        check(L.load().mpg_grid_rotang_dev(self._h, C.byref(c), C.byref(s)))
        return DevArray(c.value, self.ny * self.nx), DevArray(s.value, self.ny * self.nx)

    def rotang(self):
        """(cosalpha, sinalpha) [ny][nx] (get_rotang, model_grid.F90:2450-2507); Lambert grids only."""
        cosa, sina = np.empty((self.ny, self.nx)), np.empty((self.ny, self.nx))
        check(L.load().mpg_grid_get_rotang(self._h, _ptr(cosa), _ptr(sina)))
        return cosa, sina

    def mapfac(self, staggerloc):
        """MAPFAC_M / _U / _V (get_map_factor, model_grid.F90:2229-2365)."""
        mf = np.empty(self.stagger_shape(staggerloc))
        check(L.load().mpg_grid_get_mapfac(self._h, C.c_int(staggerloc), _ptr(mf)))
        return mf

    def stagger_shape(self, staggerloc):
        return {STAGGERLOC_CENTER: (self.ny, self.nx), STAGGERLOC_EDGE1: (self.ny, self.nx + 1),
                STAGGERLOC_EDGE2: (self.ny + 1, self.nx), STAGGERLOC_CORNER: (self.ny + 1, self.nx + 1)}[staggerloc]

    def destroy(self):
        if self._h:
            check(L.load().mpg_grid_destroy(self._h))
            self._h = C.c_void_p()


class DevArray:
    """A device array the LIBRARY owns, as far as the wrappers need one (address + element count); valid while its owner lives."""

    def __init__(self, ptr, n):
        self._ptr, self._n = int(ptr), int(n)

    def data_ptr(self):
        return self._ptr

    def numel(self):
        return self._n


class RouteHandle:
    def __init__(self, h):
        self._h = h
        n_src, n_dst, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        nx, ny, npr = C.c_int(), C.c_int(), C.c_int()
        check(L.load().mpg_handle_info(h, C.byref(n_src), C.byref(n_dst), C.byref(nx), C.byref(ny), C.byref(npr), C.byref(nnz)))
        self.n_src, self.n_dst, self.nx_dst, self.ny_dst = n_src.value, n_dst.value, nx.value, ny.value
        self.nnz_per_row, self.nnz = npr.value, nnz.value

    @property
    def store_ms(self):
        ms = C.c_float()
        check(L.load().mpg_handle_store_ms(self._h, C.byref(ms)))
        return ms.value

    @property
    def store_path(self):
        """0 hierarchical candidate search, 1 through the grid's index space, 2 index space + BVH for the rest (nearest)."""
        v = C.c_int()
        check(L.load().mpg_handle_store_path(self._h, C.byref(v)))
        return v.value

    @property
    def store_stats(self):
        """mpg_handle_store_stats: which data-dependent branches the Store took (layout by method: include/mpassit_amd.h)."""
        v = (C.c_int64 * 8)()
        check(L.load().mpg_handle_store_stats(self._h, v, C.c_int(8)))
        return [int(x) for x in v]

    def _refresh(self):
        self.__init__(self._h)

    # -- ESMF_FieldRegrid / ESMF_FieldBundleRegrid ---------------------------------------------------
    def regrid(self, src, nlev=1, nfields=1, layout=LAYOUT_CELL_FAST, out=None, src_be=False):
        """src: nfields slabs of nlev*n_src float64 (numpy on host or torch on the GPU).
        Returns dst [nfields][nlev][ny_dst][nx_dst] (squeezing nfields == 1 and nlev == 1 is left to the caller).
        src_be: the CUDA tensor holds the big-endian bytes of a NetCDF classic variable (io_nc device flow)."""
        shape = (nfields, nlev, self.ny_dst, self.nx_dst)
        need = nfields * nlev * self.n_src
        if _is_torch(src):
            import torch
            if src.is_cuda and (src.dtype == torch.float32 or src_be) and src.is_contiguous():
                # a field still in the file's NF90_FLOAT type (and byte order): widened inside the kernel's loads (same
                # result as widening first, nf90_get_var -> r8 in input_data.F90), float64 out
                return self.regrid_typed(src, nlev=nlev, nfields=nfields, layout=layout, out_dtype=torch.float64, out=out, src_be=src_be)
            if not src.is_cuda or src.dtype != torch.float64 or not src.is_contiguous():
                raise ValueError("device regrid needs a contiguous float32/float64 CUDA tensor")
            if src.numel() != need:
                raise ValueError("source has %d elements, handle expects %d" % (src.numel(), need))
            if out is None:
                out = torch.empty(shape, dtype=torch.float64, device=src.device)
            _account_regrid(self, nlev, nfields, 8, 8)
            check(L.load().mpg_regrid_dev(self._h, C.c_void_p(src.data_ptr()), C.c_int(layout), C.c_int(nlev), C.c_int(nfields),
                                          C.c_void_p(out.data_ptr()), _stream_ptr()))
            return out
        if isinstance(src, np.ndarray) and src.dtype == np.float32:
            return self.regrid_typed_host(src, nlev=nlev, nfields=nfields, layout=layout, out_dtype=np.float64, out=out)
        src = _f64(src)
        if src.size != need:
            raise ValueError("source has %d elements, handle expects %d" % (src.size, need))
        if out is None:
            out = np.empty(shape)
        check(L.load().mpg_regrid(self._h, _ptr(src), C.c_int(layout), C.c_int(nlev), C.c_int(nfields), _ptr(out)))
        return out

    def regrid_typed(self, src, nlev=1, nfields=1, layout=LAYOUT_CELL_FAST, out_dtype=None, scale=1.0, offset=0.0, out=None,
                     src_be=False, dst_be=False):
        """Fused ingest/egress Regrid on device tensors: float32 or float64 source (as in the MPAS file), float32 or
        float64 destination (as in the output file), float64 arithmetic, dst = cast(regrid(src)*scale + offset).
        src_be / dst_be: that side holds big-endian values (the bytes of a NetCDF classic variable; MPG_TYPE_BE)."""
        import torch
        if not (src.is_cuda and src.is_contiguous() and src.dtype in (torch.float32, torch.float64)):
            raise ValueError("regrid_typed needs a contiguous float32/float64 CUDA tensor")
        if src.numel() != nfields * nlev * self.n_src:
            raise ValueError("source has %d elements, handle expects %d" % (src.numel(), nfields * nlev * self.n_src))
        out_dtype = out_dtype or src.dtype
        if out is None:
            out = torch.empty((nfields, nlev, self.ny_dst, self.nx_dst), dtype=out_dtype, device=src.device)
        _account_regrid(self, nlev, nfields, src.element_size(), out.element_size())
        check(L.load().mpg_regrid_typed_dev(self._h, C.c_void_p(src.data_ptr()), C.c_int(int(src.dtype == torch.float32) | (2 if src_be else 0)),
                                            C.c_int(layout), C.c_int(nlev), C.c_int(nfields), C.c_void_p(out.data_ptr()),
                                            C.c_int(int(out.dtype == torch.float32) | (2 if dst_be else 0)), C.c_double(scale), C.c_double(offset),
                                            _stream_ptr()))
        return out

    def regrid_bundle(self, srcs, nlev=1, layout=LAYOUT_CELL_FAST, out_dtype=None, scale=1.0, offsets=None, outs=None, src_be=False, dst_be=False):
        """ESMF_FieldBundleRegrid over SEPARATE field arrays (interp.F90:240-254; mpg_regrid_bundle_typed_dev): srcs = device
        tensors of nlev * n_src elements each, one dtype; one launch for all of them, per-field epilogue offsets.  Returns the
        list of results [nlev][ny][nx] (outs: tensors to write into)."""
        import torch
        nf = len(srcs)
        if nf == 0:
            return []
        dt = srcs[0].dtype
        for t in srcs:
            if not (t.is_cuda and t.is_contiguous() and t.dtype == dt and dt in (torch.float32, torch.float64)):
                raise ValueError("regrid_bundle needs contiguous CUDA tensors of one float32 / float64 dtype")
            if t.numel() != nlev * self.n_src:
                raise ValueError("a source has %d elements, handle expects %d" % (t.numel(), nlev * self.n_src))
        out_dtype = out_dtype or dt
        if outs is None:
            outs = [torch.empty((nlev, self.ny_dst, self.nx_dst), dtype=out_dtype, device=srcs[0].device) for _ in range(nf)]
        for t in outs:
            if not (t.is_cuda and t.is_contiguous() and t.dtype == outs[0].dtype and t.numel() == nlev * self.n_dst):
                raise ValueError("regrid_bundle: bad destination tensor")
        _account_regrid(self, nlev, nf, srcs[0].element_size(), outs[0].element_size())
        sp = (C.c_void_p * nf)(*[t.data_ptr() for t in srcs])
        dp = (C.c_void_p * nf)(*[t.data_ptr() for t in outs])
        op = None if offsets is None else (C.c_double * nf)(*[float(o) for o in offsets])
        check(L.load().mpg_regrid_bundle_typed_dev(self._h, C.c_int(nf), sp, C.c_int(int(dt == torch.float32) | (2 if src_be else 0)), C.c_int(layout),
                                                   C.c_int(nlev), dp, C.c_int(int(outs[0].dtype == torch.float32) | (2 if dst_be else 0)),
                                                   C.c_double(scale), op, _stream_ptr()))
        return outs

    def regrid_typed_host(self, src, nlev=1, nfields=1, layout=LAYOUT_CELL_FAST, out_dtype=None, scale=1.0, offset=0.0, out=None):
        """The same on numpy arrays (mpg_regrid_typed): float32 / float64 host buffers cross PCIe as they are, chunks of
        levels are uploaded, regridded and downloaded concurrently.  Returns [nfields][nlev][ny][nx] of out_dtype."""
        src = np.ascontiguousarray(src)
        if src.dtype not in (np.float32, np.float64):
            src = src.astype(np.float64)
        if src.size != nfields * nlev * self.n_src:
            raise ValueError("source has %d elements, handle expects %d" % (src.size, nfields * nlev * self.n_src))
        out_dtype = np.dtype(out_dtype or src.dtype)
        if out is None:
            out = np.empty((nfields, nlev, self.ny_dst, self.nx_dst), out_dtype)
        check(L.load().mpg_regrid_typed(self._h, src.ctypes.data_as(C.c_void_p), C.c_int(int(src.dtype == np.float32)), C.c_int(layout),
                                        C.c_int(nlev), C.c_int(nfields), out.ctypes.data_as(C.c_void_p),
                                        C.c_int(int(out.dtype == np.float32)), C.c_double(scale), C.c_double(offset)))
        return out

    def regrid_bundle_host(self, srcs, nlev=1, layout=LAYOUT_CELL_FAST, out_dtype=None, scale=1.0, offsets=None, outs=None):
        """mpg_regrid_bundle_typed: the fields of a bundle as separate numpy arrays (float32 or float64, one dtype), all through
        ONE upload / Regrid / download pipeline.  Returns the list of results [nlev][ny][nx] of out_dtype."""
        nf = len(srcs)
        if nf == 0:
            return []
        srcs = [np.ascontiguousarray(a) for a in srcs]
        dt = srcs[0].dtype
        if dt not in (np.float32, np.float64) or any(a.dtype != dt for a in srcs):
            raise ValueError("regrid_bundle_host needs float32 or float64 arrays of one dtype")
        for a in srcs:
            if a.size != nlev * self.n_src:
                raise ValueError("a source has %d elements, handle expects %d" % (a.size, nlev * self.n_src))
        out_dtype = np.dtype(out_dtype or dt)
        if outs is None:
            outs = [np.empty((nlev, self.ny_dst, self.nx_dst), out_dtype) for _ in range(nf)]
        sp = (C.c_void_p * nf)(*[a.ctypes.data for a in srcs])
        dp = (C.c_void_p * nf)(*[o.ctypes.data for o in outs])
        op = None if offsets is None else (C.c_double * nf)(*[float(o) for o in offsets])
        check(L.load().mpg_regrid_bundle_typed(self._h, C.c_int(nf), sp, C.c_int(int(dt == np.float32)), C.c_int(layout), C.c_int(nlev), dp,
                                               C.c_int(int(out_dtype == np.float32)), C.c_double(scale), op))
        return outs

    @classmethod
    def from_weights(cls, n_src, nx_dst, ny_dst, row, col, S):
        """Route handle from externally computed weights in ESMF's factorList / factorIndexList form (1-based
        row = destination j*nx+i, col = source), e.g. the S/row/col of an ESMF_RegridWeightGen file."""
        row = np.ascontiguousarray(row, np.int32)
        col = np.ascontiguousarray(col, np.int32)
        S = _f64(S)
        if not (row.size == col.size == S.size):
            raise ValueError("row, col and S must have the same length")
        h = C.c_void_p()
        check(L.load().mpg_handle_from_weights(C.c_int64(n_src), C.c_int(nx_dst), C.c_int(ny_dst), C.c_int64(S.size), _ptr(row), _ptr(col),
                                               _ptr(S), C.byref(h)))
        return cls(h)

    def to_esmf_weights(self):
        """(row, col, S), 1-based, unmapped destination points omitted: what ESMF would return as factorIndexList /
        factorList for the same regrid (nearest: S = 1)."""
        if self.nnz_per_row == 0:
            rowptr, col, val = self.csr()
            row = np.repeat(np.arange(1, self.n_dst + 1, dtype=np.int32), np.diff(rowptr).astype(np.int64))
            return row, (col + 1).astype(np.int32), val
        idx, w = self.weights()
        keep = idx >= 0
        row = np.broadcast_to(np.arange(1, self.n_dst + 1, dtype=np.int32)[:, None], idx.shape)[keep]
        row, col, S = np.ascontiguousarray(row), (idx[keep] + 1).astype(np.int32), np.ascontiguousarray(w[keep])
        dst, src0, wp, row_len = self.pole()
        if len(dst):
            # a pole node's value is the mean of one CENTER row: row_len factors of w_pole / row_len each
            nz = wp != 0.0
            prow = np.repeat(dst[nz] + 1, row_len).astype(np.int32)
            pcol = (src0[nz][:, None] + np.arange(1, row_len + 1, dtype=np.int32)[None, :]).reshape(-1).astype(np.int32)
            row, col, S = np.concatenate([row, prow]), np.concatenate([col, pcol]), np.concatenate([S, np.repeat(wp[nz] / row_len, row_len)])
            keep = S != 0.0                                      # the cap points' zero-weight filler slots
            row, col, S = row[keep], col[keep], S[keep]
        return row, col, S

    def kernel_choice(self):
        """(cell_fast, lev_fast, max_unique): which Regrid kernel serves this handle (mpg_handle_kernel_choice)."""
        cf, lf, mu = C.c_int(), C.c_int(), C.c_int()
        check(L.load().mpg_handle_kernel_choice(self._h, C.byref(cf), C.byref(lf), C.byref(mu)))
        return cf.value, lf.value, mu.value

    def tile_stats(self):
        """(tile_nx, tile_ny, reuse, line_fill) of the tile lists in use (mpg_handle_tile_stats); None before the first
        staged Regrid of the handle."""
        tx, ty, reuse, fill = C.c_int(), C.c_int(), C.c_double(), C.c_double()
        if L.load().mpg_handle_tile_stats(self._h, C.byref(tx), C.byref(ty), C.byref(reuse), C.byref(fill)) != L.MPG_SUCCESS:
            return None
        return tx.value, ty.value, reuse.value, fill.value

    def pole(self):
        """Pole terms of a Grid -> Grid handle on a periodic grid: (dst_id, src_row_start, w_pole, row_len);
        empty arrays for every other handle (mpg_handle_get_pole)."""
        n, row_len = C.c_int64(), C.c_int()
        check(L.load().mpg_handle_pole_count(self._h, C.byref(n), C.byref(row_len)))
        dst, src0, wp = np.empty(n.value, np.int32), np.empty(n.value, np.int32), np.empty(n.value)
        if n.value:
            check(L.load().mpg_handle_get_pole(self._h, _ptr(dst), _ptr(src0), _ptr(wp)))
        return dst, src0, wp, row_len.value

    def weights(self):
        """(idx [n_dst][nnz_per_row] int32 with -1 = unmapped, w [n_dst][nnz_per_row])."""
        idx = np.empty((self.n_dst, self.nnz_per_row), np.int32)
        w = np.empty((self.n_dst, self.nnz_per_row))
        check(L.load().mpg_handle_get_weights(self._h, _ptr(idx), _ptr(w)))
        return idx, w

    def csr(self):
        rowptr = np.empty(self.n_dst + 1, np.int64)
        col, val = np.empty(self.nnz, np.int32), np.empty(self.nnz)
        check(L.load().mpg_handle_get_csr(self._h, _ptr(rowptr), _ptr(col), _ptr(val)))
        return rowptr, col, val

    def unique_sources(self):
        n = C.c_int64()
        check(L.load().mpg_handle_unique_sources(self._h, C.byref(n), None))
        ids = np.empty(n.value, np.int32)
        if n.value:
            check(L.load().mpg_handle_unique_sources(self._h, C.byref(n), _ptr(ids)))
        return ids

    def source_range(self):
        """(first, end): the global source ids this Mesh -> Grid handle references (mpg_handle_source_range)."""
        a, b = C.c_int64(), C.c_int64()
        check(L.load().mpg_handle_source_range(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def localize(self):
        ids = self.unique_sources()
        check(L.load().mpg_handle_localize(self._h))
        self._refresh()
        return ids

    def rebase(self, base, n_local):
        check(L.load().mpg_handle_rebase(self._h, C.c_int64(base), C.c_int64(n_local)))
        self._refresh()

    def release(self):
        if self._h:
            check(L.load().mpg_handle_release(self._h))
            self._h = None


def regrid_store(src_mesh, dst_grid, regridmethod=REGRIDMETHOD_BILINEAR, staggerloc=STAGGERLOC_CENTER,
                 meshloc=MESHLOC_ELEMENT):
    """ESMF_FieldRegridStore(mesh field -> grid field); srcTermProcessing=1, unmappedaction=IGNORE."""
    h = C.c_void_p()
    check(L.load().mpg_regrid_store(src_mesh._h, C.c_int(meshloc), dst_grid._h, C.c_int(staggerloc), C.c_int(regridmethod), C.byref(h)))
    return RouteHandle(h)


def regrid_store_grid(grid, dst_staggerloc, src_staggerloc=STAGGERLOC_CENTER, regridmethod=REGRIDMETHOD_BILINEAR):
    """ESMF_FieldRegridStore(u_target_grid_nostag -> u_target_grid) (interp.F90:298,316)."""
    h = C.c_void_p()
    check(L.load().mpg_regrid_store_grid(grid._h, C.c_int(src_staggerloc), C.c_int(dst_staggerloc), C.c_int(regridmethod), C.byref(h)))
    return RouteHandle(h)


def regrid_store_begin(src_mesh, dst_grid, regridmethod=REGRIDMETHOD_BILINEAR, staggerloc=STAGGERLOC_CENTER, meshloc=MESHLOC_ELEMENT):
    """mpg_regrid_store_begin: the Store of regrid_store(same arguments) STARTED on the library's worker thread; returns at once.  The later
    regrid_store returns the finished handle (waiting for what is left of it)."""
    check(L.load().mpg_regrid_store_begin(src_mesh._h, C.c_int(meshloc), dst_grid._h, C.c_int(staggerloc), C.c_int(regridmethod)))


def regrid_store_grid_begin(grid, dst_staggerloc, src_staggerloc=STAGGERLOC_CENTER, regridmethod=REGRIDMETHOD_BILINEAR):
    """mpg_regrid_store_grid_begin: regrid_store_grid(same arguments) started in the background."""
    check(L.load().mpg_regrid_store_grid_begin(grid._h, C.c_int(src_staggerloc), C.c_int(dst_staggerloc), C.c_int(regridmethod)))


def rotate_winds_cgrid(cosa, sina, u, v):
    """rotate_winds_cgrid (interp.F90:689-749), in place.  u, v: [nlev][ny][nx] (or [ny][nx]); numpy or torch."""
    if _is_torch(u):
        npts = cosa.numel()
        nlev = u.numel() // npts
        if ACCOUNT is not None:                 # u, v read and written once, cos / sin(alpha) read once
            ACCOUNT.append(("rotate L%d" % nlev, npts * (nlev * 32 + 16)))
        check(L.load().mpg_rotate_winds_dev(C.c_int64(npts), C.c_int(nlev), C.c_void_p(cosa.data_ptr()), C.c_void_p(sina.data_ptr()),
                                            C.c_void_p(u.data_ptr()), C.c_void_p(v.data_ptr()), _stream_ptr()))
        return u, v
    cosa, sina = _f64(cosa), _f64(sina)
    if not (u.flags.c_contiguous and v.flags.c_contiguous and u.dtype == np.float64 and v.dtype == np.float64):
        raise ValueError("u, v must be contiguous float64 (rotated in place)")
    npts = cosa.size
    nlev = u.size // npts
    check(L.load().mpg_rotate_winds(C.c_int64(npts), C.c_int(nlev), _ptr(cosa), _ptr(sina), _ptr(u), _ptr(v)))
    return u, v


def wind_destagger(rh_u, rh_v, cosa, sina, umass, vmass, nlev, out_dtype=None, dst_be=False, keep_mass=False):
    """interp.F90:291-328 in one pass (mpg_wind_destagger_dev; numpy arrays: mpg_wind_destagger): rotate_winds_cgrid on the CENTER-stagger winds (cosa / sina
    None: no rotation) + UMASS -> U(EDGE1) + VMASS -> V(EDGE2).  rh_u / rh_v: the regrid_store_grid handles of ONE grid (either
    may be None); umass / vmass: float64 CUDA tensors (or numpy arrays) [nlev][ny][nx], not modified.  Returns (U, V, UMASS', VMASS'): U
    [nlev][ny][nx+1], V [nlev][ny+1][nx] of out_dtype (float64), the rotated mass winds only with keep_mass (else None).
    Bit-identical to rotate_winds_cgrid followed by the two handles' regrid().  Raises MpgError(rc = MPG_ERR_UNSUPPORTED) for
    handles that are not such a pair (re-indexed ones)."""
    ref = umass if umass is not None else vmass
    rot = cosa is not None
    if not _is_torch(ref):
        # HOST arrays (mpg_wind_destagger): the mass winds cross the link once, U and V come back -- 2 fields up, 2 down, where
        # rotate_winds_cgrid + two regrid() calls move 4 up and 4 down.  keep_mass: the rotated mass winds are returned as new arrays.
        out_np = np.dtype(out_dtype or np.float64)
        if out_np not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError("wind_destagger: out_dtype must be float64 or float32")
        um = _f64(umass) if umass is not None else None
        vm = _f64(vmass) if vmass is not None else None
        ca, sa = (_f64(cosa), _f64(sina)) if rot else (None, None)
        u = np.empty((nlev, rh_u.ny_dst, rh_u.nx_dst), dtype=out_np) if rh_u is not None else None
        v = np.empty((nlev, rh_v.ny_dst, rh_v.nx_dst), dtype=out_np) if rh_v is not None else None
        ur = np.empty_like(um) if (keep_mass and rot) else None
        vr = np.empty_like(vm) if (keep_mass and rot) else None

        def hp(a):
            return _ptr(a) if a is not None else None
        check(L.load().mpg_wind_destagger(rh_u._h if rh_u is not None else None, rh_v._h if rh_v is not None else None, hp(ca), hp(sa), hp(um), hp(vm),
                                          C.c_int(nlev), hp(u), hp(v), C.c_int(int(out_np == np.dtype(np.float32)) | (2 if dst_be else 0)), hp(ur), hp(vr)))
        return u, v, ur, vr
    import torch
    out_dtype = out_dtype or torch.float64
    for t in (umass, vmass):
        if t is not None and not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float64):
            raise ValueError("wind_destagger needs contiguous float64 CUDA tensors")
    u = torch.empty((nlev, rh_u.ny_dst, rh_u.nx_dst), dtype=out_dtype, device=ref.device) if rh_u is not None else None
    v = torch.empty((nlev, rh_v.ny_dst, rh_v.nx_dst), dtype=out_dtype, device=ref.device) if rh_v is not None else None
    ur = torch.empty_like(umass) if (keep_mass and rot) else None
    vr = torch.empty_like(vmass) if (keep_mass and rot) else None
    if ACCOUNT is not None:     # both mass fields read once, U and V written once, indices + weights of both handles and the angles once
        npts = ref.numel() // nlev
        es = 4 if out_dtype == torch.float32 else 8
        by = 0
        if rh_u is not None:
            by += nlev * (npts * 8 + rh_u.n_dst * es) + rh_u.n_dst * 48
        if rh_v is not None:
            by += nlev * (npts * 8 + rh_v.n_dst * es) + rh_v.n_dst * 48
        ACCOUNT.append(("wind_destagger L%d" % nlev, by + (npts * 16 if rot else 0)))

    def p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None
    check(L.load().mpg_wind_destagger_dev(rh_u._h if rh_u is not None else None, rh_v._h if rh_v is not None else None, p(cosa), p(sina),
                                          p(umass), p(vmass), C.c_int(nlev), p(u), p(v),
                                          C.c_int(int(out_dtype == torch.float32) | (2 if dst_be else 0)), p(ur), p(vr), _stream_ptr()))
    return u, v, ur, vr
