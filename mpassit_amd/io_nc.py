"""File surface around the hot path: MPAS grid / history / diagnostics files in, WRF-style file out.

Mirrors the nf90 calls of the reference with the classic-format library of this repo (ncio, include/mpassit_ncio.h):
  read_grid          model_grid.F90:287-417     dimensions + lat/lon of cells and vertices, verticesOnCell, ter, zs
  read_input_data    input_data.F90:316-812     every listed variable, first Time record, file order kept:
                                                 3-D fields stay [nCells][nVertLevels] (MPG_LAYOUT_LEV_FAST), the host
                                                 transpose of :653-655 is not done
  write_target_data  write_data.F90:173-1498    dimensions, global attributes, grid variables, fields after the
                                                 post-ops of post.output_fields; NF90_FLOAT everywhere
Differences, all forced by the build image (no libnetcdf): files are NetCDF CLASSIC by default (CDF-5, CDF-2 on request:
their variables are byte ranges, which the device-resident flow moves file <-> GPU as they are); NetCDF-4 files (fmt=4 for the output; inputs
recognised by their magic) go through libhdf5 and host arrays where ncio was built with it -- a build without says how to convert.
`write_mpas_files` produces synthetic input files in the MPAS layout for tests and demos."""
import numpy as np

from . import _lib as L
from . import fields as F
from . import interp as I
from . import ncio
from . import regrid as R
from . import synth

DATESTRLEN = 19


# ---- synthetic inputs in MPAS layout -----------------------------------------------------------------------------
def write_mpas_files(grid_path, hist_path, mesh, ter, zs, hist, nz, nsoil, diag_path=None, diag=None, xtime="2024-08-07_00:00:00",
                     fmt=5, start_time=None):
    """mesh: synth.MpasMesh; hist / diag: name -> [nCells] | [nCells][L] arrays (file order), written as float32 with a
    leading Time record like MPAS history files.  Grid variables go to grid_path (static file), fields to hist_path."""
    nc, nv, me = mesh.nCells, mesh.nVertices, mesh.maxEdges
    with ncio.Writer(grid_path, format=fmt) as w:
        for name, n in (("Time", None), ("nCells", nc), ("nVertices", nv), ("maxEdges", me), ("nSoilLevels", nsoil), ("StrLen", 64)):
            w.def_dim(name, n)
        w.put_att("on_a_sphere", "YES")
        w.put_att("sphere_radius", 6371229.0)
        for name in ("latCell", "lonCell"):
            w.def_var(name, ncio.DOUBLE, ("nCells",), units="rad")
        for name in ("latVertex", "lonVertex"):
            w.def_var(name, ncio.DOUBLE, ("nVertices",), units="rad")
        w.def_var("verticesOnCell", ncio.INT, ("nCells", "maxEdges"))
        w.def_var("nEdgesOnCell", ncio.INT, ("nCells",))
        w.def_var("ter", ncio.DOUBLE, ("nCells",), units="m", long_name="terrain height")
        w.def_var("zs", ncio.FLOAT, ("Time", "nSoilLevels"), units="m")
        w.put("latCell", mesh.latCell)
        w.put("lonCell", mesh.lonCell)
        w.put("latVertex", mesh.latVertex)
        w.put("lonVertex", mesh.lonVertex)
        w.put("verticesOnCell", mesh.verticesOnCell)
        w.put("nEdgesOnCell", (mesh.verticesOnCell > 0).sum(1).astype(np.int32))
        w.put("ter", ter)
        w.put("zs", np.asarray(zs, np.float32), rec=0)

    def fields_file(path, flds):
        with ncio.Writer(path, format=fmt) as w:
            w.def_dim("Time", None)
            w.def_dim("nCells", nc)
            w.def_dim("nVertLevels", nz)
            w.def_dim("nVertLevelsP1", nz + 1)
            w.def_dim("nSoilLevels", nsoil)
            w.def_dim("StrLen", 64)
            w.put_att("config_start_time", start_time or xtime)
            w.put_att("config_dt", 20.0)
            w.put_att("config_lsm_scheme", "noah")
            w.put_att("config_microp_scheme", "mp_thompson")
            w.put_att("config_convection_scheme", "cu_grell_freitas")
            w.put_att("output_interval", np.int32(3600))
            w.def_var("xtime", ncio.CHAR, ("Time", "StrLen"))
            levdim = {nz: "nVertLevels", nz + 1: "nVertLevelsP1", nsoil: "nSoilLevels"}
            for name, a in flds.items():
                a = np.asarray(a)
                dims = ("Time", "nCells") if a.ndim == 1 else ("Time", "nCells", levdim[a.shape[1]])
                w.def_var(name, ncio.FLOAT, dims, units="-", long_name=name)
            w.put("xtime", xtime.encode().ljust(64), rec=0)
            for name, a in flds.items():
                w.put(name, np.asarray(a, np.float32), rec=0)
    fields_file(hist_path, hist)
    if diag_path is not None:
        fields_file(diag_path, diag or {})


# ---- input -----------------------------------------------------------------------------------------------------
def read_grid(path):
    """-> (synth.MpasMesh, ter [nCells] float64, zs [nsoil] float64).  model_grid.F90:287-417."""
    with ncio.Reader(path) as r:
        for d in ("nCells", "nVertices", "maxEdges"):
            if d not in r.dims:
                raise ValueError("%s: dimension %s missing -- not an MPAS grid file" % (path, d))
        mesh = synth.MpasMesh(r.get("latCell", dtype=np.float64), r.get("lonCell", dtype=np.float64),
                              r.get("latVertex", dtype=np.float64), r.get("lonVertex", dtype=np.float64),
                              np.ascontiguousarray(r.get("verticesOnCell", dtype=np.int32)))
        ter = r.get("ter", dtype=np.float64) if "ter" in r.vars else None
        zs = None
        if "zs" in r.vars:
            zs = r.get("zs", rec=0, dtype=np.float64) if r.vars["zs"]["record"] else r.get("zs", dtype=np.float64)
    return mesh, ter, zs


def _read_field(r, name):
    v = r.vars[name]
    # NF90_FLOAT variables stay single precision on the host (half the memory and PCIe bytes); the Regrid widens them in
    # its loads, which is the nf90_get_var -> r8 conversion of input_data.F90:630 done on the GPU.  Other types -> r8.
    dt = np.float32 if v["type"] == ncio.FLOAT else np.float64
    return r.get(name, rec=0, dtype=dt) if v["record"] else r.get(name, dtype=dt)


_RAW_TYPES = {ncio.FLOAT: ("float32", 4), ncio.DOUBLE: ("float64", 8)}


def bswap_(t):
    """In-place byte swap of a CUDA tensor's elements (mpg_bswap_dev): big-endian file order <-> host order."""
    import ctypes as C
    import torch
    L.check(L.load().mpg_bswap_dev(C.c_void_p(t.data_ptr()), C.c_int64(t.numel()), C.c_int(t.element_size()),
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return t


def _file_to_dev(path, offset, dev):
    """bytes [offset, offset + dev.numel()) of the file -> the 1-D uint8 CUDA tensor `dev` (mpg_file_to_dev: pinned staging
    buffers and pread threads inside the library; blocking)."""
    import ctypes as C
    import torch
    L.check(L.load().mpg_file_to_dev(str(path).encode(), C.c_int64(offset), C.c_int64(dev.numel()), C.c_void_p(dev.data_ptr()),
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)))


def _dev_to_file(path, offset, dev):
    """the 1-D uint8 CUDA tensor `dev` -> bytes [offset, ...) of the file, which already has that size (mpg_dev_to_file)."""
    import ctypes as C
    import torch
    L.check(L.load().mpg_dev_to_file(str(path).encode(), C.c_int64(offset), C.c_int64(dev.numel()), C.c_void_p(dev.data_ptr()),
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)))


def _read_field_dev(r, name, device):
    """The variable's bytes go file mapping -> GPU as stored (no host conversion pass, NF90_FLOAT stays 4 bytes on the
    bus) and STAY as stored: the tensor is marked mpg_be and Regrid reads big-endian float32 / float64 in its loads
    (MPG_TYPE_BE) -- no byte-swap pass over the data."""
    import torch
    v = r.vars[name]
    if v["type"] not in _RAW_TYPES or r.format == 4:   # (a NetCDF-4 file has no raw byte range: chunked / deflated / little-endian -- through libhdf5 and the host)
        return torch.as_tensor(_read_field(r, name), device=device)
    off, nb = r.extent(name, rec=0)
    t = torch.empty(nb, dtype=torch.uint8, device=device)
    _file_to_dev(r.path, off, t)
    shape = [int(n) for n in v["shape"]][1 if v["record"] else 0:]
    t = t.view(getattr(torch, _RAW_TYPES[v["type"]][0])).reshape(shape)
    t.mpg_be = True
    return t


_LSM = {"noah": 2, "ruc": 3}                                                  # input_data.F90:347-355
_MP = {"mp_thompson": 8, "mp_nssl2m": 18}                                       # :362-369
_CU = {"cu_ntiedke": 16, "cu_kain_fritsch": 1, "cu_grell_freitas": 3}          # :374-383


def read_file_meta(r, meta, diag=False):
    """The global attributes of an MPAS history / diag file that end up in the output header (input_data.F90:219-245 for
    the diag file, :347-389 for the history file, which is read second and wins).  Missing attributes -> the reference's
    defaults (0 / 0.0); an unknown scheme name leaves the previous value, as the reference's if-chain does."""
    def text(name):
        try:
            v = r.att(name)
        except ncio.NcioError:
            return None
        return v.strip().strip("\x00") if isinstance(v, str) else None

    def num(name):
        try:
            return float(np.atleast_1d(r.att(name))[0])
        except (ncio.NcioError, TypeError, ValueError):
            return None
    st = text("config_start_time")
    if st is not None:
        meta["start_time"] = st
    dt = num("config_dt")
    meta["config_dt"] = 0.0 if dt is None else dt
    if diag:
        oi = num("output_interval")
        meta["diag_out_interval"] = 0 if oi is None else int(oi)
        return meta
    for key, att, table in (("lsm_scheme", "config_lsm_scheme", _LSM), ("mp_scheme", "config_microp_scheme", _MP),
                            ("conv_scheme", "config_convection_scheme", _CU)):
        v = text(att)
        if v is None:
            meta[key] = 0
        elif v in table:
            meta[key] = table[v]
        else:
            meta.setdefault(key, 0)
    return meta


def xtime_seconds(start_time, valid_time):
    """datetime(start) - datetime(valid) in whole seconds, as write_target_data has it (write_data.F90:1211-1227): the
    reference subtracts in THIS order, so a valid time after the start gives a negative value; kept as written.  XTIME is
    this / 60 and ITIMESTEP int(this / config_dt) (:1233-1240) -- both from the seconds, like the reference and the Fortran
    host (seconds -> minutes -> seconds can land one ulp below a multiple of config_dt and truncate one step short)."""
    import datetime as dtm

    def parse(s):
        return dtm.datetime(int(s[0:4]), int(s[5:7]), int(s[8:10]), int(s[11:13]), int(s[14:16]), int(s[17:19]))
    try:
        return (parse(start_time) - parse(valid_time)).total_seconds()
    except ValueError:
        return 0.0


def xtime_minutes(start_time, valid_time):
    return xtime_seconds(start_time, valid_time) / 60.0


def read_input_data(hist_path, cfg, ter, diag_path=None, device=None):
    """-> (interp.InputData with layout = LEV_FAST, attrs dict name -> (units, long_name), valid time string).
    Every variable of the lists is read whole, first Time record (input_data.F90:316-812).
    device="cuda": fields become CUDA tensors of the file's own float type, moved as raw big-endian bytes and swapped
    on the GPU (device-side ingest)."""
    inp = I.InputData(hgt=ter, layout=R.LAYOUT_LEV_FAST)
    if device is not None and ter is not None:
        import torch
        inp.hgt = torch.as_tensor(np.ascontiguousarray(ter, np.float64), device=device)
    attrs = {}
    valid = ""

    def take(r, names, store):
        for n in names:
            if n not in r.vars:
                raise KeyError("variable %s not in %s" % (n, r_path))
            store[n] = _read_field(r, n) if device is None else _read_field_dev(r, n, device)
            try:
                attrs[n] = (r.att("units", var=n), r.att("long_name", var=n))
            except ncio.NcioError:
                attrs[n] = ("-", n)
    if cfg.interp_diag and diag_path is not None:
        with ncio.Reader(diag_path) as r:
            read_file_meta(r, inp.meta, diag=True)
    if cfg.interp_hist:
        r_path = hist_path
        with ncio.Reader(hist_path) as r:
            read_file_meta(r, inp.meta)
            inp.nz = r.dims.get("nVertLevels", 0)
            inp.nzp1 = r.dims.get("nVertLevelsP1", inp.nz + 1)
            inp.nsoil = r.dims.get("nSoilLevels", 0)
            take(r, [n for n, _ in cfg.hist_2d + cfg.hist_3d + cfg.hist_soil], inp.hist)
            if "xtime" in r.vars:
                valid = r.get("xtime", rec=0).tobytes().decode().strip()[:DATESTRLEN]
    if cfg.interp_diag and diag_path is not None:
        r_path = diag_path
        with ncio.Reader(diag_path) as r:
            inp.nz = inp.nz or r.dims.get("nVertLevels", 0)
            take(r, [n for n, _ in cfg.diag_list], inp.diag)
    return inp, attrs, valid


# ---- output ----------------------------------------------------------------------------------------------------
_GRID_VARS = [  # name, stagger key, dims, description, units   (write_data.F90:312-476)
    ("XLONG", "M", ("west_east", "south_north"), "LONGITUDE, WEST IS NEGATIVE", "degree_east"),
    ("XLONG_U", "U", ("west_east_stag", "south_north"), "LONGITUDE, WEST IS NEGATIVE", "degree_east"),
    ("XLONG_V", "V", ("west_east", "south_north_stag"), "LONGITUDE, WEST IS NEGATIVE", "degree_east"),
    ("XLAT", "M", ("west_east", "south_north"), "LATITUDE, SOUTH IS NEGATIVE", "degree_north"),
    ("XLAT_U", "U", ("west_east_stag", "south_north"), "LATITUDE, SOUTH IS NEGATIVE", "degree_north"),
    ("XLAT_V", "V", ("west_east", "south_north_stag"), "LATITUDE, SOUTH IS NEGATIVE", "degree_north"),
    ("MAPFAC_M", "M", ("west_east", "south_north"), "Map scale factor on mass grid", ""),
    ("MAPFAC_U", "U", ("west_east_stag", "south_north"), "Map scale factor on u-grid", ""),
    ("MAPFAC_V", "V", ("west_east", "south_north_stag"), "Map scale factor on v-grid", ""),
]


def _np(x):
    return x.cpu().numpy() if type(x).__module__.startswith("torch") else np.asarray(x)


def _is_dev(x):
    return type(x).__module__.startswith("torch") and x.is_cuda


def _put_dev(w, name, a):
    """float32 CUDA tensor -> the variable's bytes in the file mapping.  Tensors the post-ops produced big-endian
    (mpg_be) go out as they are; others are swapped on the GPU on a copy (the caller keeps its tensor)."""
    import torch
    is_be = bool(getattr(a, "mpg_be", False))
    a = a.contiguous()
    if w.format == 4:                                     # NetCDF-4: no byte range to fill; host order through libhdf5 (unwritten levels read as zeros)
        host = (bswap_(a.clone()) if is_be else a).reshape(-1).cpu().numpy()
        count = w._vars[name][2]
        if host.size < count:
            host = np.concatenate([host, np.zeros(count - host.size, host.dtype)])
        w.put(name, host, rec=0)
        return
    off, _ = w.extent(name, rec=0)                        # makes the record exist: a file being created reads as zeros
    if a.numel() and not bool(a.view(torch.int32).any()):
        return                                            # all-zero bit patterns (MU, PH, P of wrf_mod_vars): nothing to store
    be = (a if is_be else bswap_(a.clone())).view(torch.uint8).reshape(-1)
    _dev_to_file(w.path, off, be)                         # Z_C: nz of the declared nzp1 levels, the rest stays zero


def write_target_data(path, target, grid, res, cfg, nz, nzp1, nsoil, valid_time="0000-00-00_00:00:00", start_time=None,
                      zs=None, namelist=None, fmt=5, meta=None):
    """res: post.output_fields() result (float32 arrays, [L][ny][nx] C order = Fortran (i, j, L)).  grid: a
    regrid.Grid built with Grid.from_proj (coordinates / map factors / rotation come from the device).
    namelist: dict with dx, ref_lat, ref_lon, truelat1, truelat2, stand_lon for the global attributes.
    meta: what read_file_meta took from the input files (start_time, config_dt, physics schemes, diag output interval).
    Header = write_data.F90:177-560 attribute for attribute and in its order; the reference is built with -r8
    (CMakeLists.txt:80-82), so its `real` attributes are NF90_DOUBLE and so are they here."""
    meta = dict(meta or {})
    nl = dict(dx=0.0, ref_lat=0.0, ref_lon=0.0, truelat1=0.0, truelat2=0.0, stand_lon=0.0, pole_lat=90.0, pole_lon=0.0)
    nl.update(namelist or {})
    nx, ny = target.nx, target.ny
    start_time = start_time or meta.get("start_time") or valid_time
    with ncio.Writer(path, format=fmt) as w:
        for name, n in (("Time", None), ("west_east", nx), ("west_east_stag", nx + 1), ("south_north", ny), ("south_north_stag", ny + 1),
                        ("bottom_top", nz), ("bottom_top_stag", nzp1), ("soil_layers_stag", max(nsoil, 1)), ("StrLen", DATESTRLEN)):
            w.def_dim(name, n)                                                        # write_data.F90:177-194
        f8 = np.float64
        ga = [("WEST-EAST_GRID_DIMENSION", nx + 1), ("SOUTH-NORTH_GRID_DIMENSION", ny + 1), ("BOTTOM-TOP_GRID_DIMENSION", nz + 1),
              ("SIMULATION_START_DATE", start_time), ("START_DATE", start_time), ("DX", f8(nl["dx"])), ("DY", f8(nl["dx"])),
              ("DT", f8(meta.get("config_dt", 0.0))), ("SF_SURFACE_PHYSICS", int(meta.get("lsm_scheme", 0))),
              ("MP_PHYSICS", int(meta.get("mp_scheme", 0))), ("CU_PHYSICS", int(meta.get("conv_scheme", 0))),
              ("CEN_LAT", f8(nl["ref_lat"])), ("CEN_LON", f8(nl["ref_lon"])), ("TRUELAT1", f8(nl["truelat1"])),
              ("TRUELAT2", f8(nl["truelat2"])), ("MOAD_CEN_LAT", f8(nl["ref_lat"])), ("STAND_LON", f8(nl["stand_lon"])),
              ("POLE_LAT", f8(nl["pole_lat"])), ("POLE_LON", f8(nl["pole_lon"])), ("POL_ELAT", f8(nl["pole_lat"])),   # sic, :253
              ("MAP_PROJ", int(target.proj.code)),
              ("MAP_PROJ_CHAR", target.extra.get("map_proj_char") or
               {1: "Lambert Conformal", 2: "Polar Stereographic", 3: "Mercator"}.get(int(target.proj.code), "Lat/Lon"))]   # program_setup.F90:169-187
        if cfg.interp_diag:
            ga.append(("PREC_ACC_DT", int(meta.get("diag_out_interval", 0))))                                          # :262-265
        ga += [("I_PARENT_START", 1), ("J_PARENT_START", 1)]
        ga += [("%s_PATCH_START_%s" % (d, st), 1) for d in ("WEST-EAST", "SOUTH-NORTH", "BOTTOM-TOP") for st in ("UNSTAG", "STAG")]
        ga += [("WEST-EAST_PATCH_END_UNSTAG", nx), ("WEST-EAST_PATCH_END_STAG", nx + 1), ("SOUTH-NORTH_PATCH_END_UNSTAG", ny),
               ("SOUTH-NORTH_PATCH_END_STAG", ny + 1), ("BOTTOM-TOP_PATCH_END_UNSTAG", nz), ("BOTTOM-TOP_PATCH_END_STAG", nz + 1)]
        for k, v in ga:                                                               # :196-308
            w.put_att(k, v)
        stag = {"M": (R.STAGGERLOC_CENTER, ""), "U": (R.STAGGERLOC_EDGE1, "X"), "V": (R.STAGGERLOC_EDGE2, "Y")}
        for name, st, dims, desc, units in _GRID_VARS:
            w.def_var(name, ncio.FLOAT, ("Time",) + dims[::-1], description=desc, units=units, MemoryOrder="XY ", stagger=stag[st][1],
                      FieldType=104)
        lc = target.proj.code == 1
        if lc:                                                                        # :446-476
            for name in ("SINALPHA", "COSALPHA"):
                w.def_var(name, ncio.FLOAT, ("Time", "south_north", "west_east"), description="Local %s of map rotation" %
                          ("sine" if name[0] == "S" else "cosine"), units="", MemoryOrder="XY ", stagger="", FieldType=104)
        w.def_var("ZS", ncio.FLOAT, ("Time", "soil_layers_stag"), description="DEPTHS OF CENTERS OF SOIL LAYERS", units="m", MemoryOrder="X")
        w.def_var("Times", ncio.CHAR, ("Time", "StrLen"), description="Times", units="m", coordinates="Time", stagger="", FieldType=104)
        w.def_var("ITIMESTEP", ncio.INT, ("Time",), description="", units="", stagger="", FieldType=106, MemoryOrder="O ")     # :537-548
        w.def_var("XTIME", ncio.FLOAT, ("Time",), description="minutes since " + start_time, units="minutes since " + start_time,
                  stagger="", FieldType=104, MemoryOrder="O ")                                                              # :550-561
        dimsets = {(ny, nx): ("south_north", "west_east"), (nz, ny, nx): ("bottom_top", "south_north", "west_east"),
                   (nz, ny, nx + 1): ("bottom_top", "south_north", "west_east_stag"),
                   (nz, ny + 1, nx): ("bottom_top", "south_north_stag", "west_east"),
                   (nzp1, ny, nx): ("bottom_top_stag", "south_north", "west_east")}
        if nsoil and nsoil not in (nz, nzp1):
            dimsets[(nsoil, ny, nx)] = ("soil_layers_stag", "south_north", "west_east")
        arrays = {}
        for name, a in res.items():
            if name == "P_TOP":
                w.def_var("P_TOP", ncio.FLOAT, ("Time",), description="PRESSURE TOP OF THE MODEL", units="Pa")
                continue
            if not (_is_dev(a) and str(a.dtype) == "torch.float32"):
                a = _np(a)
            arrays[name] = a
            dims = dimsets[tuple(a.shape)]
            if name == "Z_C":                                                         # declared on bottom_top_stag, nz levels written (:479, :1413)
                dims = ("bottom_top_stag",) + dims[1:]
            w.def_var(name, ncio.FLOAT, ("Time",) + dims, MemoryOrder="XYZ" if a.ndim == 3 else "XY ", coordinates="XLONG XLAT XTIME",
                      stagger="X" if a.shape[-1] == nx + 1 else ("Y" if a.shape[-2] == ny + 1 else ""), FieldType=104)
        # ---- data (:1003-1475) ----
        from_file = bool(target.extra.get("from_file"))        # target_grid_type = 'file': the grid variables are the file's own
        host = {"M": (target.lon, target.lat, "mapfac_m"), "U": (target.lon_u, target.lat_u, "mapfac_u"), "V": (target.lon_v, target.lat_v, "mapfac_v")}
        for name, st, dims, _, _ in _GRID_VARS:
            if from_file:
                lon, lat, mk = host[st]
                mf = target.extra.get(mk)
                mf = np.ones_like(lon) if mf is None else mf
            else:
                lon, lat = grid.coords(stag[st][0])
                mf = grid.mapfac(stag[st][0]) if name.startswith("MAPFAC") else None
            w.put(name, mf if name.startswith("MAPFAC") else (lon if "LONG" in name else lat), rec=0)
        if lc:
            cosa, sina = (target.cosa, target.sina) if from_file else grid.rotang()
            w.put("SINALPHA", sina, rec=0)
            w.put("COSALPHA", cosa, rec=0)
        w.put("ZS", np.zeros(max(nsoil, 1), np.float32) if zs is None else np.asarray(zs, np.float32), rec=0)
        w.put("Times", valid_time.encode()[:DATESTRLEN].ljust(DATESTRLEN), rec=0)
        xs = xtime_seconds(start_time, valid_time)                                                                          # :1211-1240
        dt_cfg = float(meta.get("config_dt", 0.0))
        w.put("XTIME", np.array([xs / 60.0], np.float32), rec=0)
        w.put("ITIMESTEP", np.array([int(xs / dt_cfg) if dt_cfg > 0.0 else 0], np.int32), rec=0)
        for name, a in arrays.items():
            if _is_dev(a):
                _put_dev(w, name, a)
                continue
            if name == "Z_C":
                pad = np.zeros((nzp1,) + a.shape[1:], np.float32)
                pad[:nz] = a
                a = pad
            w.put(name, a, rec=0)
        if "P_TOP" in res:
            w.put("P_TOP", np.array([res["P_TOP"]], np.float32), rec=0)


def run(grid_path, hist_path, out_path, target, cfg, diag_path=None, namelist=None, fmt=5, device_io=False, timings=None):
    """mpassit.F90's main sequence on files: read grid + data, regrid on the GPU, post-ops, write.  Returns the
    dictionary that went to the file (float32, post.output_fields order; with device_io the CUDA tensors hold the values
    big-endian, as the file does).
    device_io: variables travel file <-> GPU as raw big-endian bytes and every field stays device resident in between
    (same file, bit for bit, as the host-array path).  timings: dict that receives the wall seconds of each stage."""
    import time
    from . import post

    def lap(key, t0):
        if timings is not None:
            if device_io:
                import torch
                torch.cuda.synchronize()
            timings[key] = timings.get(key, 0.0) + time.perf_counter() - t0
        return time.perf_counter()
    t = time.perf_counter()
    mpas, ter, zs = read_grid(grid_path)
    t = lap("read_grid", t)
    inp, _, valid = read_input_data(hist_path, cfg, ter, diag_path=diag_path, device="cuda" if device_io else None)
    t = lap("read_input_data", t)
    mesh = R.Mesh.from_mpas(mpas)
    grid = R.Grid.from_target(target) if target.extra.get("from_file") else R.Grid.from_proj(target)
    t = lap("mesh_grid_create", t)
    out = I.interp_data(mesh, grid, target, inp, cfg)
    t = lap("interp_data", t)
    res = post.output_fields(out, cfg, be=device_io)     # device flow: float32 results as the file stores them
    t = lap("post_ops", t)
    write_target_data(out_path, target, grid, res, cfg, inp.nz, inp.nzp1, inp.nsoil, valid_time=valid or "0000-00-00_00:00:00", zs=zs,
                      namelist=namelist, fmt=fmt, meta=inp.meta)
    t = lap("write_target_data", t)
    mesh.destroy()
    grid.destroy()
    return res


def run_series(grid_path, jobs, target, cfg, namelist=None, fmt=5, timings=None):
    """A run of time levels on one mesh and one target grid (one forecast: one history / diag file pair per output time).
    jobs: list of (hist_path, diag_path or None, out_path).  The reference is started once per file and redoes everything;
    here the mesh, the target grid and every weight set are built once, one time level of interp_data is captured into a
    hipGraph (interp.GraphedInterp) and replayed per file, variables travel file <-> GPU as raw bytes (device_io), and a
    reader thread fetches the next file's variables on its own stream while the current level is regridded and written.
    Returns the number of files written.  timings: dict that receives setup_s and the wall seconds of every file."""
    import time

    import torch
    from . import post
    if not jobs:
        return 0
    t0 = time.perf_counter()
    mpas, ter, zs = read_grid(grid_path)
    mesh = R.Mesh.from_mpas(mpas)
    grid = R.Grid.from_target(target) if target.extra.get("from_file") else R.Grid.from_proj(target)
    stream = torch.cuda.Stream()

    def fetch(job):
        with torch.cuda.stream(stream):
            got = read_input_data(job[0], cfg, ter, diag_path=job[1], device="cuda")
            stream.synchronize()
        return got
    from concurrent.futures import ThreadPoolExecutor
    reader = ThreadPoolExecutor(1)
    nxt = reader.submit(fetch, jobs[0])
    inp, _, valid = nxt.result()
    gi = I.GraphedInterp(mesh, grid, target, inp, cfg)           # Stores + eager pass + capture, on the first file's fields
    if timings is not None:
        timings["setup_s"] = time.perf_counter() - t0
        timings["files_s"] = []
    try:
        for k, job in enumerate(jobs):
            t0 = time.perf_counter()
            if k > 0:
                new, _, valid = nxt.result()
                for store, fresh in ((inp.hist, new.hist), (inp.diag, new.diag)):
                    for name, buf in store.items():
                        buf.copy_(fresh[name])                   # into the graph's static input buffers
                        fresh[name].record_stream(torch.cuda.current_stream())   # allocated on the reader's stream
                inp.meta = new.meta                              # every file carries its own start time / time step / schemes
                del new
            if k + 1 < len(jobs):
                nxt = reader.submit(fetch, jobs[k + 1])
            out = gi.replay()
            res = post.output_fields(out, cfg, be=True)
            write_target_data(job[2], target, grid, res, cfg, inp.nz, inp.nzp1, inp.nsoil, valid_time=valid or "0000-00-00_00:00:00",
                              zs=zs, namelist=namelist, fmt=fmt, meta=inp.meta)
            if timings is not None:
                torch.cuda.synchronize()
                timings["files_s"].append(time.perf_counter() - t0)
    finally:
        reader.shutdown(wait=True)
        gi.close()
        mesh.destroy()
        grid.destroy()
    return len(jobs)


__all__ = ["write_mpas_files", "read_grid", "read_input_data", "read_file_meta", "xtime_minutes", "write_target_data", "run", "run_series", "F"]
