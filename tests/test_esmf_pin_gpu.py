"""ESMF pin kit with the library on the GPU (tools/esmf_pin.py compare / selftest): the six handles of an exported case are built
from the exported files, the library's own weights survive the round trip through ESMF's weight-file layout with ZERO difference
(regional and global grid: the periodic EDGE1 column mapping and the pole rows' expansion included), and a weight file that did
NOT come from the library -- the oracle's weights standing in for the file ESMF_RegridWeightGen would write -- is compared within
the north-star tolerances, line type reported, ESMF-side weights applied through mpg_handle_from_weights."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import mesh_xyz

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("esmf_pin_tool", os.path.join(ROOT, "tools", "esmf_pin.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("workload", ["tiny", "c5_small"])
def test_own_weights_survive_the_weight_file(gpu_lib, tmp_path, workload):
    from types import SimpleNamespace

    from mpassit_amd import esmf_pin as E, workloads
    T = _tool()
    m, g, _, _ = workloads.workload(workload)
    E.export_case(str(tmp_path), m, g, name=workload)
    T.cmd_selftest(SimpleNamespace(dir=str(tmp_path), verbose=False))      # raises SystemExit on any difference
    assert len([f for f in os.listdir(str(tmp_path)) if f.startswith("selftest_weights_")]) == 6


def test_a_foreign_weight_file_is_compared_within_the_north_star_tolerances(gpu_lib, oracle, tmp_path):
    from mpassit_amd import esmf_pin as E, workloads
    T = _tool()
    m, g, _, _ = workloads.workload("tiny")
    d = str(tmp_path)
    E.export_case(d, m, g, name="tiny")
    cxyz, vxyz = mesh_xyz(oracle, m)
    pxyz = oracle.lonlat_deg_to_xyz(g.lon, g.lat)
    n_a, n_b = m.nCells, g.nx * g.ny
    # the stand-in for ESMF's files: the oracle's weights, in the layout and numbering ESMF_RegridWeightGen uses
    tri, _ = oracle.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    idx, w = oracle.bilinear_weights(cxyz, tri, pxyz)
    keep = idx >= 0
    E.write_weight_file(os.path.join(d, "weights_bilinear_center.nc"), np.broadcast_to(np.arange(1, n_b + 1)[:, None], idx.shape)[keep], idx[keep] + 1, w[keep],
                        n_a, n_b, "bilinear")
    near = oracle.nearest(cxyz, pxyz)
    E.write_weight_file(os.path.join(d, "weights_neareststod_center.nc"), np.arange(1, n_b + 1), near + 1, np.ones(n_b), n_a, n_b, "neareststod")
    rp, col, val = oracle.conserve(m.verticesOnCell, vxyz, g.nx, g.ny, oracle.lonlat_deg_to_xyz(g.lon_c, g.lat_c))
    E.write_weight_file(os.path.join(d, "weights_conserve_center.nc"), np.repeat(np.arange(1, n_b + 1), np.diff(rp)), col + 1, val, n_a, n_b, "conserve")
    cs = T._Case(d)
    for case, tol in (("bilinear_center", 1e-10), ("neareststod_center", 0.0), ("conserve_center", 1e-9)):
        rep = T._compare_case(cs, case, os.path.join(d, cs.info["cases"][case]["weights"]), verbose=True)
        assert rep["ok"] and rep["max_abs_diff_any"] <= tol and rep["mapped_only_ours"] == rep["mapped_only_theirs"] == 0, rep
        assert rep["field_rel_diff"] <= 1e-12                            # the foreign weights through mpg_handle_from_weights + Regrid
    # the line-type question answered from the file: the oracle's normal-projection weights are recognised as linetype 1
    idx1, w1 = oracle.bilinear_weights(cxyz, tri, pxyz, linetype=1)
    keep = idx1 >= 0
    E.write_weight_file(os.path.join(d, "weights_bilinear_center.nc"), np.broadcast_to(np.arange(1, n_b + 1)[:, None], idx1.shape)[keep], idx1[keep] + 1,
                        w1[keep], n_a, n_b, "bilinear")
    rep = T._compare_case(cs, "bilinear_center", os.path.join(d, "weights_bilinear_center.nc"), verbose=False)
    assert rep["ok"] and rep["max_abs_diff_any"] <= 1e-10 and rep["closest_setting"] == {"linetype": 1}
    # round 6: the other two open choices are answered from the files the same way.  A stand-in "ESMF" that fans every polygon of the
    # node-located case from its LAST vertex is recognised as node_fan_origin = -1 ...
    ftri, _ = oracle.fan_triangles(m.verticesOnCell, vxyz, origin=-1)
    idxn, wn = oracle.bilinear_weights(vxyz, ftri, pxyz)
    keep = idxn >= 0
    E.write_weight_file(os.path.join(d, "weights_bilinear_node.nc"), np.broadcast_to(np.arange(1, n_b + 1)[:, None], idxn.shape)[keep], idxn[keep] + 1,
                        wn[keep], m.nVertices, n_b, "bilinear")
    rep = T._compare_case(cs, "bilinear_node", os.path.join(d, "weights_bilinear_node.nc"), verbose=True)
    assert rep["ok"] and rep["max_abs_diff_any"] <= 1e-10 and rep["closest_setting"] == {"linetype": 0, "fan_origin": -1}, rep["closest_setting"]
    # ... and one whose Grid -> Grid Store takes a stagger point as inside within 1e-4 of a quad as grid_inside_tol_exp = 4 (on this coarse
    # grid the mapped masks of the settings differ at the hull only where a point sits within the tolerance of it: the default must at
    # least compare no better than the setting the file was made with)
    cen = oracle.lonlat_deg_to_xyz(g.lon, g.lat)
    oi, ow = oracle.grid_bilinear(g.nx, g.ny, cen, 2, oracle.lonlat_deg_to_xyz(g.lon_v, g.lat_v), tol=1e-4)
    keep = oi >= 0
    nv = g.lon_v.size
    E.write_weight_file(os.path.join(d, "weights_bilinear_center_to_edge2.nc"), np.broadcast_to(np.arange(1, nv + 1)[:, None], oi.shape)[keep], oi[keep] + 1,
                        ow[keep], g.nx * g.ny, nv, "bilinear")
    rep = T._compare_case(cs, "bilinear_center_to_edge2", os.path.join(d, "weights_bilinear_center_to_edge2.nc"), verbose=True)
    assert rep["ok"] and rep["max_abs_diff_any"] <= 1e-10 and rep["only_ours"] == rep["only_theirs"] == 0
    # a file for another grid is refused, a missing one skipped with exit code 2
    E.write_weight_file(os.path.join(d, "weights_conserve_center.nc"), [1], [1], [1.0], n_a + 1, n_b, "conserve")
    with pytest.raises(SystemExit, match="sources"):
        T._compare_case(cs, "conserve_center", os.path.join(d, "weights_conserve_center.nc"), verbose=False)
    cs.mesh.destroy()
    cs.grid.destroy()


def test_weight_files_made_from_the_50_digit_goldens_compare_clean(gpu_lib, tmp_path):
    """A stand-in for ESMF's files that shares no code with the library OR the oracle: the brute-force 50-digit Stores of
    tests/golden/store_hp.json (case 1: regional mesh, rim, unmapped points) written in ESMF_RegridWeightGen's layout and put through
    `compare` -- the whole kit, files to verdict, against an independent answer."""
    from mpassit_amd import esmf_pin as E, target_grid as tg
    from test_store_goldens import cases
    T = _tool()
    case = cases()[0]
    g = tg.define_target_grid_params("lambert", 18, 14, dx=120000.0, dy=120000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5,
                                     stand_lon=-97.5)
    assert np.array_equal(g.lon, case.lon) and np.array_equal(g.lat_c, case.lat_c)
    d = str(tmp_path)
    E.export_case(d, case.mesh, g, name="golden")
    n_a, n_b = case.mesh.nCells, case.nx * case.ny
    row, col, S = [], [], []
    for p, e in enumerate(case.expect["bilinear"]):
        if e is not None:
            assert "tie" not in e
            row += [p + 1] * 3
            col += [c + 1 for c in e["col"]]
            S += e["w"]
    E.write_weight_file(os.path.join(d, "weights_bilinear_center.nc"), row, col, S, n_a, n_b, "bilinear")
    E.write_weight_file(os.path.join(d, "weights_neareststod_center.nc"), np.arange(1, n_b + 1), [e["col"] + 1 for e in case.expect["nearest"]], np.ones(n_b),
                        n_a, n_b, "neareststod")
    cons = case.expect["conserve"]
    E.write_weight_file(os.path.join(d, "weights_conserve_center.nc"), [r + 1 for r, _, _ in cons], [c + 1 for _, c, _ in cons], [v for _, _, v in cons],
                        n_a, n_b, "conserve")
    cs = T._Case(d)
    for name in ("bilinear_center", "neareststod_center", "conserve_center"):
        rep = T._compare_case(cs, name, os.path.join(d, cs.info["cases"][name]["weights"]), verbose=True)
        assert rep["ok"] and rep["max_abs_diff_any"] <= 2e-13 and rep["mapped_only_ours"] == rep["mapped_only_theirs"] == 0, rep
    cs.mesh.destroy()
    cs.grid.destroy()
