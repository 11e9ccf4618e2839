"""ESMF pin kit: the files a site WITH ESMF needs to pin this library's weights to ESMF's own -- and the comparison.

Parity of this repository is "unpinned at the ESMF boundary": every interpolation call of the reference lands in ESMF
(interp.F90:123-136, 207-447; CMakeLists.txt:46-48), and neither ESMF nor fixtures of its results exist where this was built.
This module closes the loop from the other side.  It writes

  * the source mesh as an ESMF unstructured-mesh file (ESMFMESH: nodeCoords / elementConn / numElementConn / centerCoords in
    degrees exactly as model_grid.F90:446-497 hands them to ESMF_MeshCreate -- nodes = MPAS vertices, elements = MPAS cells,
    connectivity = the non-zero entries of verticesOnCell in file order, longitudes wrapped to (-180, 180]);
  * the target grid as SCRIP grid files (grid_center_* = the stagger's points, grid_corner_* = the CORNER stagger's points around
    each CENTER cell, counter-clockwise from (i, j): model_grid.F90:784-794) -- one per stagger the reference regrids onto;
  * a shell script with the ESMF_RegridWeightGen invocations that correspond to the reference's ESMF_FieldRegridStore calls
    (unmappedaction = IGNORE -> `-i`; srcTermProcessing is a property of the route handle's sparse multiply, not of the weights),

and it reads an ESMF weight file (`S`, `row`, `col`: 1-based destination / source sequence indices) and compares it with a weight
set of this library in the same triplet form (regrid.RouteHandle.to_esmf_weights): entries on one side only, the largest
difference on the common ones, destination points mapped on one side only -- against the tolerances BASELINE.json's north_star
states (nearest: identical; bilinear / conservative: 1e-6).

numpy + the classic-NetCDF writer only (mpassit_amd/ncio.py: CDF-2 "64-bit offset" files, which every netCDF library and ESMF
read); nothing here touches the GPU.  tools/esmf_pin.py is the command line on top (its `compare` builds the library's handle).
File layouts follow the ESMF Reference Manual, "File Formats" (SCRIP Grid File Format, ESMF Unstructured Grid File Format,
Regrid Interpolation Weight File Format)."""
import json
import os

import numpy as np

from . import ncio

# ---- names the ESMF manual prescribes (the structural test checks the written files against these) -------------------------
ESMFMESH_DIMS = ("nodeCount", "elementCount", "maxNodePElement", "coordDim")
ESMFMESH_VARS = {"nodeCoords": ("nodeCount", "coordDim"), "elementConn": ("elementCount", "maxNodePElement"),
                 "numElementConn": ("elementCount",), "centerCoords": ("elementCount", "coordDim")}
SCRIP_DIMS = ("grid_size", "grid_corners", "grid_rank")
SCRIP_VARS = {"grid_dims": ("grid_rank",), "grid_center_lat": ("grid_size",), "grid_center_lon": ("grid_size",), "grid_imask": ("grid_size",),
              "grid_corner_lat": ("grid_size", "grid_corners"), "grid_corner_lon": ("grid_size", "grid_corners")}
WEIGHT_DIMS = ("n_a", "n_b", "n_s")
WEIGHT_VARS = {"col": ("n_s",), "row": ("n_s",), "S": ("n_s",)}
ESMF_METHOD_NAMES = {"bilinear": "Bilinear", "conserve": "First-order Conservative", "neareststod": "Nearest source to destination"}

TOLERANCE = {"bilinear": 1e-6, "conserve": 1e-6, "neareststod": 0.0}   # BASELINE.json north_star


def mesh_coords_deg(lon_rad, lat_rad):
    """Radians of the MPAS file -> the degrees the reference gives ESMF (model_grid.F90:450-454, 464-468: x * 180 / pi with
    pi = 4 atan(1), lon > 180 -> lon - 360); the same two operations in the same order as csrc/k_setup.hip k_mesh_coords."""
    lon = np.asarray(lon_rad, np.float64) * 180.0 / np.pi
    lon = np.where(lon > 180.0, lon - 360.0, lon)
    return lon, np.asarray(lat_rad, np.float64) * 180.0 / np.pi


def write_esmf_mesh(path, mesh):
    """mesh: anything with latCell, lonCell, latVertex, lonVertex (radians) and verticesOnCell [nCells][maxEdges] (1-based, 0 =
    padding), as read from an MPAS grid file (model_grid.F90:341-349)."""
    voc = np.ascontiguousarray(mesh.verticesOnCell, np.int32)
    n_cells, max_edges = voc.shape
    nlon, nlat = mesh_coords_deg(mesh.lonVertex, mesh.latVertex)
    clon, clat = mesh_coords_deg(mesh.lonCell, mesh.latCell)
    # model_grid.F90:474-485: the element's nodes are the non-zero entries of its row, in file order (compacted to the front)
    nz = voc > 0
    num = nz.sum(axis=1)
    order = np.argsort(~nz, axis=1, kind="stable")
    conn = np.take_along_axis(voc, order, axis=1)
    conn[np.arange(max_edges)[None, :] >= num[:, None]] = -1
    with ncio.Writer(path, format=2) as w:
        w.def_dim("nodeCount", int(nlon.size))
        w.def_dim("elementCount", int(n_cells))
        w.def_dim("maxNodePElement", int(max_edges))
        w.def_dim("coordDim", 2)
        w.def_var("nodeCoords", ncio.DOUBLE, ("nodeCount", "coordDim"), units="degrees")
        w.def_var("elementConn", ncio.INT, ("elementCount", "maxNodePElement"), long_name="Node Indices that define the element connectivity",
                  _FillValue=np.array([-1], np.int32), start_index=np.array([1], np.int32))
        w.def_var("numElementConn", ncio.BYTE, ("elementCount",), long_name="Number of nodes per element")
        w.def_var("centerCoords", ncio.DOUBLE, ("elementCount", "coordDim"), units="degrees")
        w.put_att("gridType", "unstructured")
        w.put_att("version", "0.9")
        w.put_att("title", "MPAS mesh as the reference hands it to ESMF_MeshCreate (model_grid.F90:446-497): nodes = vertices, elements = cells")
        w.put("nodeCoords", np.stack([nlon, nlat], axis=1))
        w.put("elementConn", conn)
        w.put("numElementConn", num.astype(np.int8))
        w.put("centerCoords", np.stack([clon, clat], axis=1))
    return dict(nodeCount=int(nlon.size), elementCount=int(n_cells), maxNodePElement=int(max_edges))


def write_mpas_arrays(path, mesh):
    """The five MPAS arrays themselves (radians, as in the grid file): what `compare` hands to mpg_mesh_create, so that the library
    computes its degrees from the same bits ESMF's file was written from."""
    voc = np.ascontiguousarray(mesh.verticesOnCell, np.int32)
    with ncio.Writer(path, format=2) as w:
        w.def_dim("nCells", voc.shape[0])
        w.def_dim("nVertices", int(np.asarray(mesh.latVertex).size))
        w.def_dim("maxEdges", voc.shape[1])
        for name, dims in (("latCell", ("nCells",)), ("lonCell", ("nCells",)), ("latVertex", ("nVertices",)), ("lonVertex", ("nVertices",))):
            w.def_var(name, ncio.DOUBLE, dims, units="rad")
        w.def_var("verticesOnCell", ncio.INT, ("nCells", "maxEdges"))
        for name in ("latCell", "lonCell", "latVertex", "lonVertex"):
            w.put(name, np.asarray(getattr(mesh, name), np.float64))
        w.put("verticesOnCell", voc)


def read_mpas_arrays(path):
    from types import SimpleNamespace
    with ncio.Reader(path) as r:
        m = SimpleNamespace(**{k: r.get(k) for k in ("latCell", "lonCell", "latVertex", "lonVertex", "verticesOnCell")})
    m.nCells, m.nVertices = int(m.latCell.size), int(m.latVertex.size)
    return m


def _cell_corners(lon_c, lat_c):
    """[ny+1][nx+1] CORNER-stagger points -> [ny*nx][4] corners of every CENTER cell, counter-clockwise from (i, j):
    (i, j), (i+1, j), (i+1, j+1), (i, j+1) -- the cell ESMF builds from the CORNER stagger (model_grid.F90:784-794, 959-984)."""
    def four(a):
        return np.stack([a[:-1, :-1], a[:-1, 1:], a[1:, 1:], a[1:, :-1]], axis=-1).reshape(-1, 4)
    return four(np.asarray(lon_c, np.float64)), four(np.asarray(lat_c, np.float64))


def write_scrip_grid(path, lon, lat, lon_c=None, lat_c=None, title="target grid"):
    """lon / lat: [nj][ni] points of one stagger (degrees).  lon_c / lat_c: [nj+1][ni+1] corner points (the CORNER stagger for
    the CENTER stagger: the conservative destination cells); without them every cell's corners collapse onto its point, which is
    all a non-conservative method reads."""
    lon, lat = np.asarray(lon, np.float64), np.asarray(lat, np.float64)
    nj, ni = lon.shape
    if lon_c is not None:
        clon, clat = _cell_corners(lon_c, lat_c)
    else:
        clon, clat = np.repeat(lon.reshape(-1, 1), 4, axis=1), np.repeat(lat.reshape(-1, 1), 4, axis=1)
    with ncio.Writer(path, format=2) as w:
        w.def_dim("grid_size", ni * nj)
        w.def_dim("grid_corners", 4)
        w.def_dim("grid_rank", 2)
        w.def_var("grid_dims", ncio.INT, ("grid_rank",))
        w.def_var("grid_center_lat", ncio.DOUBLE, ("grid_size",), units="degrees")
        w.def_var("grid_center_lon", ncio.DOUBLE, ("grid_size",), units="degrees")
        w.def_var("grid_imask", ncio.INT, ("grid_size",), units="unitless")
        w.def_var("grid_corner_lat", ncio.DOUBLE, ("grid_size", "grid_corners"), units="degrees")
        w.def_var("grid_corner_lon", ncio.DOUBLE, ("grid_size", "grid_corners"), units="degrees")
        w.put_att("title", title)
        w.put("grid_dims", np.array([ni, nj], np.int32))          # fastest dimension first (SCRIP)
        w.put("grid_center_lat", lat.reshape(-1))
        w.put("grid_center_lon", lon.reshape(-1))
        w.put("grid_imask", np.ones(ni * nj, np.int32))
        w.put("grid_corner_lat", clat)
        w.put("grid_corner_lon", clon)
    return dict(grid_size=ni * nj, grid_dims=(ni, nj))


def read_scrip_grid(path):
    """-> dict(ni, nj, lon [nj][ni], lat, corner_lon [nj*ni][4], corner_lat)."""
    with ncio.Reader(path) as r:
        ni, nj = (int(x) for x in r.get("grid_dims"))
        return dict(ni=ni, nj=nj, lon=r.get("grid_center_lon").reshape(nj, ni), lat=r.get("grid_center_lat").reshape(nj, ni),
                    corner_lon=r.get("grid_corner_lon"), corner_lat=r.get("grid_corner_lat"))


def corners_from_scrip(sg):
    """The [nj+1][ni+1] CORNER-stagger arrays back from a SCRIP file written by write_scrip_grid (corner 0 of every cell, the last
    column from corner 1, the last row from corners 3 and 2)."""
    ni, nj = sg["ni"], sg["nj"]
    out = []
    for a in (sg["corner_lon"], sg["corner_lat"]):
        a = a.reshape(nj, ni, 4)
        c = np.empty((nj + 1, ni + 1))
        c[:nj, :ni] = a[:, :, 0]
        c[:nj, ni] = a[:, -1, 1]
        c[nj, :ni] = a[-1, :, 3]
        c[nj, ni] = a[-1, -1, 2]
        out.append(c)
    return out


def write_weight_file(path, row, col, S, n_a, n_b, method="bilinear", title="weights of mpassit_amd in ESMF_RegridWeightGen's layout", fmt=2):
    """(row, col, S) in ESMF's convention -- 1-based, row = destination, col = source sequence index -- as the weight file
    ESMF_RegridWeightGen writes (the `--weight_only` subset: col, row, S and the three dimensions; the attributes it sets)."""
    row, col, S = np.asarray(row, np.int32), np.asarray(col, np.int32), np.asarray(S, np.float64)
    assert row.shape == col.shape == S.shape and row.ndim == 1
    with ncio.Writer(path, format=fmt) as w:       # fmt 2 = --64bit_offset; 4 = --netcdf4 (needs the HDF5 backend of ncio)
        w.def_dim("n_a", int(n_a))
        w.def_dim("n_b", int(n_b))
        w.def_dim("n_s", max(int(S.size), 1))
        w.def_var("col", ncio.INT, ("n_s",))
        w.def_var("row", ncio.INT, ("n_s",))
        w.def_var("S", ncio.DOUBLE, ("n_s",))
        w.put_att("title", title)
        w.put_att("normalization", "destarea")
        w.put_att("map_method", "Conservative remapping" if method == "conserve" else "Bilinear remapping")
        w.put_att("ESMF_regrid_method", ESMF_METHOD_NAMES[method])
        w.put_att("conventions", "NCAR-CSM")
        w.put_att("n_s_valid", np.array([S.size], np.int32))
        if S.size:
            w.put("col", col)
            w.put("row", row)
            w.put("S", S)
        else:                                                     # (a dimension of length 0 would be the unlimited one)
            w.put("col", np.zeros(1, np.int32))
            w.put("row", np.zeros(1, np.int32))
            w.put("S", np.zeros(1))


def read_weight_file(path):
    """-> dict(row, col, S, n_a, n_b, method): an ESMF_RegridWeightGen file (full or --weight_only) in classic / 64-bit-offset /
    CDF-5 form, or NetCDF-4 (the generator's --netcdf4) where libmpassit_ncio has its HDF5 backend (otherwise: `nccopy -k cdf5 in.nc
    out.nc` first, or run the generator with --64bit_offset, as the exported script does)."""
    with ncio.Reader(path) as r:
        for v in ("S", "row", "col"):
            if v not in r.vars:
                raise ValueError("%s: no variable %r -- not an ESMF weight file" % (path, v))
        row, col, S = r.get("row").astype(np.int64), r.get("col").astype(np.int64), r.get("S").astype(np.float64)
        n_a, n_b = r.dims.get("n_a"), r.dims.get("n_b")
        method = None
        try:
            method = r.att("ESMF_regrid_method")
        except Exception:      # noqa: BLE001 -- an optional attribute
            pass
        try:
            nv = r.att("n_s_valid")
            if not isinstance(nv, str) and int(nv[0]) == 0:
                row, col, S = row[:0], col[:0], S[:0]
        except Exception:      # noqa: BLE001
            pass
    return dict(row=row, col=col, S=S, n_a=n_a, n_b=n_b, method=method)


def _keyed(row, col, S, n_a):
    """Sum duplicate (row, col) entries (ESMF may list one more than once: the pole's contributions), drop exact zeros."""
    row, col, S = np.asarray(row, np.int64), np.asarray(col, np.int64), np.asarray(S, np.float64)
    key = (row - 1) * int(n_a) + (col - 1)
    o = np.argsort(key, kind="stable")
    key, S = key[o], S[o]
    uk, start = np.unique(key, return_index=True)
    val = np.add.reduceat(S, start) if S.size else S
    keep = val != 0.0
    return uk[keep], val[keep]


def compare_weights(ours, theirs, n_a, n_b, method="bilinear", sliver=1e-12):
    """ours / theirs: (row, col, S) triplets, 1-based.  Returns a report dict; report["ok"] says whether the two weight sets agree
    within the north-star tolerance of `method`:
      nearest      identical (row, col) sets;
      bilinear     per destination point the two sparse rows agree within 1e-6 as vectors over the source ids -- so a point on a shared
                   edge that the two sides gave to different triangles (the same value either way) counts as what it is;
      conservative common entries within 1e-6, entries on one side only below 1e-6 (overlap slivers).
    Also reported: the number of entries on one side only and their largest weight, the largest difference on the common entries,
    destination points mapped on one side only, the largest row-sum difference."""
    ko, vo = _keyed(*ours, n_a)
    kt, vt = _keyed(*theirs, n_a)
    common, io, it = np.intersect1d(ko, kt, assume_unique=True, return_indices=True)
    only_o = np.ones(ko.size, bool)
    only_o[io] = False
    only_t = np.ones(kt.size, bool)
    only_t[it] = False
    d_common = np.abs(vo[io] - vt[it])
    rows_o, rows_t = np.unique(ko // int(n_a)), np.unique(kt // int(n_a))
    mapped_only_o, mapped_only_t = np.setdiff1d(rows_o, rows_t), np.setdiff1d(rows_t, rows_o)
    # per-destination-row difference as sparse vectors: the union of the keys, missing entries = 0
    allk = np.union1d(ko, kt)
    a, b = np.zeros(allk.size), np.zeros(allk.size)
    a[np.searchsorted(allk, ko)] = vo
    b[np.searchsorted(allk, kt)] = vt
    row_of = allk // int(n_a)
    rowdiff_max = float(np.abs(a - b).max()) if allk.size else 0.0
    rs_o, rs_t = np.bincount(row_of, weights=a, minlength=int(n_b)), np.bincount(row_of, weights=b, minlength=int(n_b))
    tol = TOLERANCE[method]
    rep = dict(method=method, n_a=int(n_a), n_b=int(n_b), nnz_ours=int(ko.size), nnz_theirs=int(kt.size), common=int(common.size),
               only_ours=int(only_o.sum()), only_theirs=int(only_t.sum()),
               only_ours_max=float(vo[only_o].max()) if only_o.any() else 0.0, only_theirs_max=float(np.abs(vt[only_t]).max()) if only_t.any() else 0.0,
               max_abs_diff_common=float(d_common.max()) if d_common.size else 0.0,
               max_abs_diff_any=rowdiff_max, max_rowsum_diff=float(np.abs(rs_o - rs_t).max()) if allk.size else 0.0,
               mapped_only_ours=int(mapped_only_o.size), mapped_only_theirs=int(mapped_only_t.size), tolerance=tol)
    if d_common.size:
        k = int(d_common.argmax())
        rep["worst_common_entry"] = dict(row=int(common[k] // int(n_a)) + 1, col=int(common[k] % int(n_a)) + 1, ours=float(vo[io][k]), theirs=float(vt[it][k]))
    if method == "neareststod":
        rep["ok"] = rep["only_ours"] == 0 and rep["only_theirs"] == 0
    else:
        rep["ok"] = rowdiff_max <= tol and rep["mapped_only_ours"] == 0 and rep["mapped_only_theirs"] == 0
        # destination points on the hull's rim may be mapped on one side only (a point within rounding of the last triangle's edge):
        # reported, and tolerated only when every such row's entries are what a rim point carries -- this tool cannot know, so it fails
    rep["sliver_entries_one_side"] = int((vo[only_o] < sliver).sum() + (np.abs(vt[only_t]) < sliver).sum())
    return rep


def report_text(rep, label_ours="mpassit_amd", label_theirs="ESMF"):
    L = ["method %s: %d sources, %d destination points" % (rep["method"], rep["n_a"], rep["n_b"]),
         "  entries: %s %d, %s %d, common %d" % (label_ours, rep["nnz_ours"], label_theirs, rep["nnz_theirs"], rep["common"]),
         "  (row, col) only in %s: %d (largest weight %.3e); only in %s: %d (largest %.3e)" % (
             label_ours, rep["only_ours"], rep["only_ours_max"], label_theirs, rep["only_theirs"], rep["only_theirs_max"]),
         "  max |dS| on common entries %.3e; as sparse rows (missing = 0) %.3e; max row-sum difference %.3e" % (
             rep["max_abs_diff_common"], rep["max_abs_diff_any"], rep["max_rowsum_diff"]),
         "  destination points mapped only by %s: %d, only by %s: %d" % (label_ours, rep["mapped_only_ours"], label_theirs, rep["mapped_only_theirs"])]
    if "worst_common_entry" in rep:
        w = rep["worst_common_entry"]
        L.append("  worst common entry: row %d col %d: %.17g vs %.17g" % (w["row"], w["col"], w["ours"], w["theirs"]))
    L.append("  tolerance %.1e -> %s" % (rep["tolerance"], "WITHIN TOLERANCE" if rep["ok"] else "OUT OF TOLERANCE"))
    return "\n".join(L)


# ---- the export of one (mesh, target grid) pair ---------------------------------------------------------------------------------
# The reference's Store calls (interp.F90) -> what asks ESMF_RegridWeightGen for the same weights.  `-i` = unmappedaction IGNORE
# (interp.F90:127); the line type is left at ESMF's default, as the reference leaves it (`-l cartesian` / `-l greatcircle` select it).
CASES = [
    # key, method, source file key, --src_loc, destination file key, reference call site
    ("bilinear_center", "bilinear", "mesh", "center", "grid_center", "interp.F90:123,207,226,241,259,277,334 (elements -> CENTER)"),
    ("conserve_center", "conserve", "mesh", None, "grid_center", "interp.F90:372,394 (snow, snowh)"),
    ("neareststod_center", "neareststod", "mesh", "center", "grid_center", "interp.F90:421 (ivgtyp, isltyp, xland, landmask)"),
    ("bilinear_node", "bilinear", "mesh", "corner", "grid_center", "interp.F90:353 (vorticity: values on the mesh's nodes)"),
    ("bilinear_center_to_edge1", "bilinear", "grid_center", None, "grid_edge1", "interp.F90:298 (UMASS -> U)"),
    ("bilinear_center_to_edge2", "bilinear", "grid_center", None, "grid_edge2", "interp.F90:316 (VMASS -> V)"),
]


def export_case(out_dir, mesh, target, name="case"):
    """Writes <out_dir>/{src_mesh_esmf.nc, src_mesh_mpas.nc, dst_grid_center_scrip.nc, dst_grid_edge1_scrip.nc,
    dst_grid_edge2_scrip.nc, run_esmf_regridweightgen.sh, manifest.json}.  target: target_grid.TargetGrid with its arrays."""
    os.makedirs(out_dir, exist_ok=True)
    files = dict(mesh="src_mesh_esmf.nc", mpas="src_mesh_mpas.nc", grid_center="dst_grid_center_scrip.nc", grid_edge1="dst_grid_edge1_scrip.nc",
                 grid_edge2="dst_grid_edge2_scrip.nc")
    info = dict(name=name, files=files, regional=bool(target.is_regional), nx=int(target.nx), ny=int(target.ny))
    info["mesh"] = write_esmf_mesh(os.path.join(out_dir, files["mesh"]), mesh)
    write_mpas_arrays(os.path.join(out_dir, files["mpas"]), mesh)
    info["grid_center"] = write_scrip_grid(os.path.join(out_dir, files["grid_center"]), target.lon, target.lat, target.lon_c, target.lat_c,
                                           title="CENTER stagger (mass points) with the CORNER stagger's cells")
    # a periodic grid's EDGE1 stagger holds the first nx columns only in ESMF (the duplicate column one period later is ours)
    lon_u, lat_u = (target.lon_u, target.lat_u) if target.is_regional else (target.lon_u[:, :-1], target.lat_u[:, :-1])
    info["grid_edge1"] = write_scrip_grid(os.path.join(out_dir, files["grid_edge1"]), lon_u, lat_u, title="EDGE1 stagger (U points)")
    info["grid_edge2"] = write_scrip_grid(os.path.join(out_dir, files["grid_edge2"]), target.lon_v, target.lat_v, title="EDGE2 stagger (V points)")
    lines = ["#!/bin/sh", "# ESMF_RegridWeightGen runs that ask ESMF for the weights of the reference's ESMF_FieldRegridStore calls (interp.F90).",
             "# Needs ESMF >= 8.3 (CMakeLists.txt:48).  --64bit_offset keeps the weight files in a classic format tools/esmf_pin.py reads;",
             "# -i = ESMF_UNMAPPEDACTION_IGNORE (interp.F90:127).  Then: python tools/esmf_pin.py compare <dir> [--case KEY]", "set -e",
             "RWG=${ESMF_REGRIDWEIGHTGEN:-ESMF_RegridWeightGen}", "MPIRUN=${MPIRUN:-}", ""]
    regional = bool(target.is_regional)
    for key, method, src, src_loc, dst, site in CASES:
        a = ["$MPIRUN $RWG", "-s", files[src], "-d", files[dst], "-w", "weights_%s.nc" % key, "-m", method, "-i", "--64bit_offset"]
        a += ["--src_type", "ESMF" if src == "mesh" else "SCRIP", "--dst_type", "SCRIP"]
        if src_loc:
            a += ["--src_loc", src_loc]
        if src != "mesh" and regional:
            a += ["--src_regional"]
        if regional:
            a += ["--dst_regional"]
        if src != "mesh" and not regional:
            a += ["-p", "all"]      # ESMF_POLEMETHOD_ALLAVG, the default the reference leaves in force on its 1PeriDim / MONOPOLE grid (model_grid.F90:685-694)
        lines += ["# %s -- %s" % (key, site), " ".join(a), ""]
    sh = os.path.join(out_dir, "run_esmf_regridweightgen.sh")
    with open(sh, "w") as f:
        f.write("\n".join(lines))
    os.chmod(sh, 0o755)
    info["cases"] = {key: dict(method=method, src=src, src_loc=src_loc, dst=dst, weights="weights_%s.nc" % key, reference=site)
                     for key, method, src, src_loc, dst, site in CASES}
    with open(os.path.join(out_dir, "manifest.json"), "w") as f:
        json.dump(info, f, indent=1)
    return info
