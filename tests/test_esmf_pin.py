"""ESMF pin kit on the CPU (mpassit_amd/esmf_pin.py, tools/esmf_pin.py export): the exported files have the structure the ESMF
Reference Manual prescribes for an unstructured-mesh file, a SCRIP grid file and a regrid weight file, they hold what the reference
hands to ESMF (model_grid.F90:446-497, 784-794), and a weight set written in ESMF's layout and read back compares as identical
to itself -- while every kind of disagreement the comparison is there to find (a weight off by more than the tolerance, an entry
on one side only, a nearest neighbour that differs, a destination point mapped on one side only) is found.  The "theirs" side here is
the oracle's weights (this image has no ESMF): test infrastructure standing in for the file ESMF_RegridWeightGen would write."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import mesh_xyz

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# names from the ESMF Reference Manual ("ESMF Unstructured Grid File Format", "SCRIP Grid File Format", "Regrid Interpolation Weight
# File Format"), written out here independently of the module's own constants
MANUAL_ESMFMESH = {"dims": {"nodeCount", "elementCount", "maxNodePElement", "coordDim"},
                   "vars": {"nodeCoords": ("nodeCount", "coordDim"), "elementConn": ("elementCount", "maxNodePElement"),
                            "numElementConn": ("elementCount",), "centerCoords": ("elementCount", "coordDim")}}
MANUAL_SCRIP = {"dims": {"grid_size", "grid_corners", "grid_rank"},
                "vars": {"grid_dims": ("grid_rank",), "grid_center_lat": ("grid_size",), "grid_center_lon": ("grid_size",),
                         "grid_imask": ("grid_size",), "grid_corner_lat": ("grid_size", "grid_corners"), "grid_corner_lon": ("grid_size", "grid_corners")}}
MANUAL_WEIGHTS = {"dims": {"n_a", "n_b", "n_s"}, "vars": {"col": ("n_s",), "row": ("n_s",), "S": ("n_s",)}}


@pytest.fixture(scope="module")
def exported(tmp_path_factory):
    from mpassit_amd import workloads
    d = str(tmp_path_factory.mktemp("pin"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "esmf_pin.py"), "export", "tiny", "--out", d], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    m, g, _, _ = workloads.workload("tiny")
    return d, m, g


def _structure(path, manual):
    from mpassit_amd import ncio
    with ncio.Reader(path) as r:
        assert r.format in (1, 2, 5)                                    # a classic format every netCDF library (and ESMF) reads
        assert set(r.dims) >= manual["dims"], (set(r.dims), manual["dims"])
        for v, dims in manual["vars"].items():
            assert v in r.vars and r.vars[v]["dims"] == dims, (v, r.vars.get(v))


def test_mesh_file_is_what_the_reference_gives_esmf(exported, oracle):
    from mpassit_amd import esmf_pin as E, ncio
    d, m, g = exported
    path = os.path.join(d, "src_mesh_esmf.nc")
    _structure(path, MANUAL_ESMFMESH)
    assert E.ESMFMESH_VARS == MANUAL_ESMFMESH["vars"]
    with ncio.Reader(path) as r:
        assert r.att("gridType") == "unstructured" and r.att("units", "nodeCoords") == "degrees" and r.att("units", "centerCoords") == "degrees"
        assert int(r.att("_FillValue", "elementConn")[0]) == -1 and r.vars["numElementConn"]["type"] in (ncio.BYTE, ncio.INT)
        assert r.dims["coordDim"] == 2 and r.dims["nodeCount"] == m.nVertices and r.dims["elementCount"] == m.nCells
        node, cen, conn, num = r.get("nodeCoords"), r.get("centerCoords"), r.get("elementConn"), r.get("numElementConn").astype(np.int64)
    # degrees exactly as model_grid.F90:450-454,464-468 computes them == the oracle's restatement == what the library computes
    lon_d, lat_d = oracle.mesh_coords_deg(m.lonCell, m.latCell)
    vlon_d, vlat_d = oracle.mesh_coords_deg(m.lonVertex, m.latVertex)
    assert np.array_equal(cen[:, 0], lon_d) and np.array_equal(cen[:, 1], lat_d)
    assert np.array_equal(node[:, 0], vlon_d) and np.array_equal(node[:, 1], vlat_d)
    assert node[:, 0].max() <= 180.0 and node[:, 0].min() > -180.0 - 1e-12
    # connectivity: the non-zero entries of verticesOnCell in file order (model_grid.F90:448,474-485), 1-based, padded with -1
    assert np.array_equal(num, (m.verticesOnCell > 0).sum(axis=1))
    for c in (0, 1, m.nCells // 2, m.nCells - 1):
        row = m.verticesOnCell[c]
        assert np.array_equal(conn[c, :num[c]], row[row > 0]) and (conn[c, num[c]:] == -1).all()
    assert conn[conn > 0].max() <= m.nVertices and (conn != 0).all()
    # the radians themselves travel beside the ESMF file, for the library's side of the comparison
    back = E.read_mpas_arrays(os.path.join(d, "src_mesh_mpas.nc"))
    for k in ("latCell", "lonCell", "latVertex", "lonVertex", "verticesOnCell"):
        assert np.array_equal(getattr(back, k), getattr(m, k))


def test_scrip_files_hold_the_staggers(exported):
    from mpassit_amd import esmf_pin as E
    d, m, g = exported
    for f in ("dst_grid_center_scrip.nc", "dst_grid_edge1_scrip.nc", "dst_grid_edge2_scrip.nc"):
        _structure(os.path.join(d, f), MANUAL_SCRIP)
    c = E.read_scrip_grid(os.path.join(d, "dst_grid_center_scrip.nc"))
    assert (c["ni"], c["nj"]) == (g.nx, g.ny)                              # grid_dims: fastest dimension first
    assert np.array_equal(c["lon"], g.lon) and np.array_equal(c["lat"], g.lat)
    lon_c, lat_c = E.corners_from_scrip(c)
    assert np.array_equal(lon_c, g.lon_c) and np.array_equal(lat_c, g.lat_c)   # the CORNER stagger round-trips through the cells' corners
    # corners counter-clockwise (SCRIP's rule): positive shoelace area in the lon / lat plane of this mid-latitude regional grid
    x, y = c["corner_lon"], c["corner_lat"]
    area = 0.5 * sum(x[:, k] * y[:, (k + 1) % 4] - x[:, (k + 1) % 4] * y[:, k] for k in range(4))
    assert (area > 0).all()
    u, v = E.read_scrip_grid(os.path.join(d, "dst_grid_edge1_scrip.nc")), E.read_scrip_grid(os.path.join(d, "dst_grid_edge2_scrip.nc"))
    assert np.array_equal(u["lon"], g.lon_u) and np.array_equal(v["lat"], g.lat_v) and (u["ni"], v["nj"]) == (g.nx + 1, g.ny + 1)


def test_script_asks_for_the_reference_calls(exported):
    d, m, g = exported
    sh = open(os.path.join(d, "run_esmf_regridweightgen.sh")).read()
    runs = [ln for ln in sh.splitlines() if ln.startswith("$MPIRUN $RWG")]
    assert len(runs) == 6
    assert all(" -i " in r and "--64bit_offset" in r and "--dst_regional" in r for r in runs)      # unmappedaction = IGNORE (interp.F90:127)
    by = {r.split("-w weights_")[1].split(".nc")[0]: r for r in runs}
    assert "-m bilinear" in by["bilinear_center"] and "--src_loc center" in by["bilinear_center"] and "--src_type ESMF" in by["bilinear_center"]
    assert "-m conserve" in by["conserve_center"] and "--src_loc" not in by["conserve_center"]
    assert "-m neareststod" in by["neareststod_center"]
    assert "--src_loc corner" in by["bilinear_node"]                       # vorticity: values on the mesh's nodes (interp.F90:353)
    for k, dst in (("bilinear_center_to_edge1", "edge1"), ("bilinear_center_to_edge2", "edge2")):
        assert "-s dst_grid_center_scrip.nc" in by[k] and "-d dst_grid_%s_scrip.nc" % dst in by[k] and "--src_regional" in by[k] and "--src_type SCRIP" in by[k]


def test_exported_files_open_in_an_independent_netcdf_reader(exported, tmp_path):
    """ESMF reads its inputs through libnetcdf; the nearest thing here is scipy's own classic-NetCDF reader: every exported file (and a
    weight file) opens there as a 64-bit-offset classic file with the same dimensions, attributes and numbers."""
    from scipy.io import netcdf_file

    from mpassit_amd import esmf_pin as E
    d, m, g = exported
    with netcdf_file(os.path.join(d, "src_mesh_esmf.nc"), "r", mmap=False) as f:
        assert f.version_byte == 2 and f.dimensions["nodeCount"] == m.nVertices and f.dimensions["elementCount"] == m.nCells and f.dimensions["coordDim"] == 2
        assert f.gridType == b"unstructured" and f.variables["nodeCoords"].units == b"degrees"
        conn, num = f.variables["elementConn"][:], f.variables["numElementConn"][:]
        assert conn.shape == m.verticesOnCell.shape and conn.dtype.kind == "i" and num.dtype.itemsize == 1 and int(f.variables["elementConn"]._FillValue) == -1
        assert np.array_equal(num, (m.verticesOnCell > 0).sum(axis=1)) and conn.min() >= -1 and (conn != 0).all() and conn.max() == m.nVertices
        lon = np.asarray(m.lonCell) * 180.0 / np.pi
        assert np.array_equal(f.variables["centerCoords"][:, 0], np.where(lon > 180.0, lon - 360.0, lon))
    with netcdf_file(os.path.join(d, "dst_grid_center_scrip.nc"), "r", mmap=False) as f:
        assert tuple(f.variables["grid_dims"][:]) == (g.nx, g.ny) and f.dimensions["grid_size"] == g.nx * g.ny and f.dimensions["grid_corners"] == 4
        assert np.array_equal(f.variables["grid_center_lat"][:], g.lat.reshape(-1)) and f.variables["grid_center_lon"].units == b"degrees"
        assert np.array_equal(f.variables["grid_corner_lon"][:, 2], g.lon_c[1:, 1:].reshape(-1)) and (f.variables["grid_imask"][:] == 1).all()
    w = str(tmp_path / "w.nc")
    E.write_weight_file(w, [3, 1], [2, 7], [0.25, 1.0], 9, 4, "conserve")
    with netcdf_file(w, "r", mmap=False) as f:
        assert (f.dimensions["n_a"], f.dimensions["n_b"], f.dimensions["n_s"]) == (9, 4, 2)
        assert list(f.variables["row"][:]) == [3, 1] and list(f.variables["col"][:]) == [2, 7] and list(f.variables["S"][:]) == [0.25, 1.0]
        assert f.ESMF_regrid_method == b"First-order Conservative" and f.normalization == b"destarea"


def _oracle_triplets(oracle, m, g, method):
    cxyz, vxyz = mesh_xyz(oracle, m)
    pxyz = oracle.lonlat_deg_to_xyz(g.lon, g.lat)
    if method == "conserve":
        rp, col, val = oracle.conserve(m.verticesOnCell, vxyz, g.nx, g.ny, oracle.lonlat_deg_to_xyz(g.lon_c, g.lat_c))
        row = np.repeat(np.arange(1, g.nx * g.ny + 1), np.diff(rp))
        return row, col + 1, val
    if method == "neareststod":
        idx = oracle.nearest(cxyz, pxyz)
        return np.arange(1, idx.size + 1), idx + 1, np.ones(idx.size)
    tri, _ = oracle.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    idx, w = oracle.bilinear_weights(cxyz, tri, pxyz)
    keep = idx >= 0
    row = np.broadcast_to(np.arange(1, idx.shape[0] + 1)[:, None], idx.shape)[keep]
    return row, idx[keep] + 1, w[keep]


@pytest.mark.parametrize("method", ["bilinear", "conserve", "neareststod"])
def test_weight_file_round_trip_and_what_the_comparison_finds(exported, oracle, method, tmp_path):
    from mpassit_amd import esmf_pin as E
    d, m, g = exported
    n_a, n_b = m.nCells, g.nx * g.ny
    row, col, S = _oracle_triplets(oracle, m, g, method)
    path = str(tmp_path / "w.nc")
    E.write_weight_file(path, row, col, S, n_a, n_b, method)
    _structure(path, MANUAL_WEIGHTS)
    back = E.read_weight_file(path)
    assert back["n_a"] == n_a and back["n_b"] == n_b and back["method"] == E.ESMF_METHOD_NAMES[method]
    # entry order does not matter (ESMF's files are not sorted): shuffle one side
    perm = np.random.default_rng(1).permutation(S.size)
    rep = E.compare_weights((row[perm], col[perm], S[perm]), (back["row"], back["col"], back["S"]), n_a, n_b, method)
    assert rep["ok"] and rep["only_ours"] == rep["only_theirs"] == 0 and rep["max_abs_diff_any"] == 0.0 and rep["common"] == S.size
    assert "WITHIN TOLERANCE" in E.report_text(rep)
    k = S.size // 2
    if method != "neareststod":
        # below the tolerance: reported, accepted; above it: refused
        S2 = S.copy()
        S2[k] += 3e-9
        rep = E.compare_weights((row, col, S2), (row, col, S), n_a, n_b, method)
        assert rep["ok"] and 2e-9 < rep["max_abs_diff_common"] < 4e-9 and rep["worst_common_entry"]["row"] == row[k]
        S2[k] += 1e-5
        rep = E.compare_weights((row, col, S2), (row, col, S), n_a, n_b, method)
        assert not rep["ok"] and "OUT OF TOLERANCE" in E.report_text(rep)
        # an entry on one side only: a sliver is tolerated, a real weight is not
        big = int(np.argmax(S))
        rep = E.compare_weights((np.delete(row, big), np.delete(col, big), np.delete(S, big)), (row, col, S), n_a, n_b, method)
        assert not rep["ok"] and rep["only_theirs"] == 1 and rep["only_theirs_max"] == S[big]
        rep = E.compare_weights((np.append(row, row[k]), np.append(col, (col[k] % n_a) + 1 if (col[k] % n_a) + 1 not in col[row == row[k]] else 1), np.append(S, 1e-13)),
                                (row, col, S), n_a, n_b, method)
        assert rep["ok"] and rep["only_ours"] == 1 and rep["sliver_entries_one_side"] == 1
        # a destination point mapped on one side only
        r0 = row[k]
        drop = row == r0
        rep = E.compare_weights((row[~drop], col[~drop], S[~drop]), (row, col, S), n_a, n_b, method)
        assert not rep["ok"] and rep["mapped_only_theirs"] == 1
    else:
        col2 = col.copy()
        col2[k] = col[k] % n_a + 1                                          # another nearest neighbour for one point
        rep = E.compare_weights((row, col2, S), (row, col, S), n_a, n_b, method)
        assert not rep["ok"] and rep["only_ours"] == 1 and rep["only_theirs"] == 1


def test_duplicate_entries_are_summed_and_empty_sets_compare(tmp_path):
    """ESMF lists a pole's contributions as several factors for one (row, col): they count as their sum."""
    from mpassit_amd import esmf_pin as E
    ours = (np.array([1, 2]), np.array([3, 1]), np.array([0.75, 1.0]))
    theirs = (np.array([2, 1, 1]), np.array([1, 3, 3]), np.array([1.0, 0.5, 0.25]))
    rep = E.compare_weights(ours, theirs, 4, 2, "bilinear")
    assert rep["ok"] and rep["common"] == 2 and rep["max_abs_diff_any"] == 0.0
    path = str(tmp_path / "e.nc")
    E.write_weight_file(path, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0), 4, 2, "conserve")
    back = E.read_weight_file(path)
    assert back["S"].size == 0
    assert E.compare_weights((back["row"], back["col"], back["S"]), (back["row"], back["col"], back["S"]), 4, 2, "conserve")["ok"]


def test_global_grid_drops_the_duplicate_edge1_column(tmp_path):
    """A periodic grid's EDGE1 stagger has nx columns in ESMF (ours carries the duplicate one period later); no regional flags, the
    default pole method spelled out for the Grid -> Grid runs."""
    from mpassit_amd import esmf_pin as E, workloads
    m, g, _, _ = workloads.workload("c5_small")
    info = E.export_case(str(tmp_path), m, g, name="c5_small")
    assert info["grid_edge1"]["grid_dims"] == (g.nx, g.ny) and info["grid_edge2"]["grid_dims"] == (g.nx, g.ny + 1) and not info["regional"]
    sh = open(os.path.join(str(tmp_path), "run_esmf_regridweightgen.sh")).read()
    assert "_regional" not in sh and sh.count("-p all") == 2


def test_a_netcdf4_weight_file_is_read_too(tmp_path):
    """ESMF_RegridWeightGen --netcdf4 writes its weights into an HDF5 container: read like the classic ones where ncio has HDF5."""
    from mpassit_amd import esmf_pin as E, ncio
    if not ncio.has_netcdf4():
        pytest.skip("libmpassit_ncio was built without HDF5")
    rng = np.random.default_rng(2)
    key = rng.choice(50 * 80, 200, replace=False)                       # distinct (row, col) entries
    row, col, S = key // 80 + 1, key % 80 + 1, rng.uniform(0, 1, 200)
    for fmt in (2, 4):
        E.write_weight_file(tmp_path / ("w%d.nc" % fmt), row, col, S, 80, 50, "conserve", fmt=fmt)
    a, b = E.read_weight_file(tmp_path / "w2.nc"), E.read_weight_file(tmp_path / "w4.nc")
    assert open(tmp_path / "w4.nc", "rb").read(4) == b"\x89HDF" and a["method"] == b["method"] and (a["n_a"], a["n_b"]) == (b["n_a"], b["n_b"]) == (80, 50)
    for k in ("row", "col", "S"):
        assert np.array_equal(a[k], b[k])
    rep = E.compare_weights((a["row"], a["col"], a["S"]), (b["row"], b["col"], b["S"]), 80, 50, "conserve")
    assert rep["ok"] and rep["max_abs_diff_any"] == 0.0
