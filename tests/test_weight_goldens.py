"""The oracle's two weight formulas against tests/golden/weights_hp.json: 160 (triangle, point) pairs and 30 (polygon, 4 x 4
grid) cases from 3-km to 500-km cells, evaluated by tests/golden/make_weight_goldens.py along a different route at 50 digits
(3 x 3 solve instead of the determinant closed form; Girard's angle excess instead of the triangle fan).  Inputs are the
stored float64 values exactly, so what is compared is the double-precision arithmetic of the oracle alone; the kernels are
tied to the oracle by the parity tests (tests/test_parity_gpu.py, tests/test_configs_gpu.py).  Not a pin on ESMF."""
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "weights_hp.json")))


@pytest.mark.parametrize("linetype", [0, 1])
def test_bilinear_triangle_weights(oracle, linetype):
    """linetype 0: the point meets the triangle's plane along the ray from the centre (golden `w`); 1: along the plane's
    normal (golden `w_normal`) -- the two readings of straight cell edges on a sphere, DESIGN.md s2."""
    worst = {}
    for c in GOLD["bilinear"]:
        tri = np.array(c["tri"])
        idx, w = oracle.bilinear_weights(tri, np.array([[0, 1, 2]], np.int32), np.array([c["p"]]), linetype=linetype)
        assert sorted(idx[0].tolist()) == [0, 1, 2], c
        got = np.empty(3)
        got[idx[0]] = w[0]                                   # weight of vertex k, whatever order the triangle is stored in
        err = np.abs(got - np.array(c["w_normal" if linetype else "w"])).max()
        worst[c["scale_km"]] = max(worst.get(c["scale_km"], 0.0), err)
        assert abs(got.sum() - 1) < 4e-16
    # the difference form keeps km-scale triangles as accurate as continental ones
    assert max(worst.values()) < 2e-13, worst


def test_conservative_overlap_weights(oracle):
    worst = {}
    for c in GOLD["conserve"]:
        src, cor = np.array(c["src"]), np.array(c["corners"])
        nv = src.shape[0]
        voc = np.zeros((1, 8), np.int32)
        voc[0, :nv] = np.arange(1, nv + 1)
        rp, col, val = oracle.conserve(voc, src, 4, 4, cor)
        got = np.zeros(16)
        for p in range(16):
            if rp[p + 1] > rp[p]:
                assert rp[p + 1] - rp[p] == 1 and col[rp[p]] == 0
                got[p] = val[rp[p]]
        want = np.array(c["w"])
        sliver = (want < 1e-12) | (got == 0.0) & (want < 1e-11)      # the code drops overlaps below 1e-14 of the cell
        err = np.abs(got - want)[~sliver].max(initial=0.0)
        key = (c["scale_km"], c["ratio"])
        worst[key] = max(worst.get(key, 0.0), err)
        assert (got[sliver] < 1e-11).all()
        assert want.max() > 0.01                                     # the case does overlap
    # unit vectors carry 1e-16 of absolute position: a cell of angular size h has its area to ~1e-16 / h relative
    for (scale_km, ratio), e in worst.items():
        h = scale_km * ratio / 6370.0
        assert e < 64 * 2.2e-16 / h, (scale_km, ratio, e, worst)


def test_quad_bilinear_weights(oracle):
    """Grid -> grid bilinear (CENTER -> EDGE, the destaggering of U and V): four weights per point on a 2 x 2 CENTER grid."""
    worst = {}
    for c in GOLD["quad"]:
        idx, w = oracle.grid_bilinear(2, 2, np.array(c["centres"]), 1, np.array(c["pts"]))
        for p in (1, 4):      # the U points between the two CENTER columns: the ones whose candidate quads include the only quad
            assert sorted(idx[p].tolist()) == [0, 1, 2, 3], (c["scale_km"], idx[p])
            got = np.empty(4)
            got[idx[p]] = w[p]
            err = np.abs(got - np.array(c["w"][p])).max()
            worst[c["scale_km"]] = max(worst.get(c["scale_km"], 0.0), err)
    assert max(worst.values()) < 1e-12, worst


# ---- the same known answers through the C-ABI (GPU) ------------------------------------------------------------------------
# The kernels take latitude / longitude (radians for the mesh, degrees for the grid), not unit vectors: the goldens' float64
# unit vectors go through asin / atan2 here and through the device's sin / cos there, two or three roundings of 1e-16 rad each,
# i.e. up to ~1e-12 of a 3-km cell -- the tolerance below is 5e-12.
def _latlon_rad(xyz):
    xyz = np.asarray(xyz, np.float64)
    return np.arcsin(np.clip(xyz[:, 2], -1, 1)), np.arctan2(xyz[:, 1], xyz[:, 0])


@pytest.mark.gpu
@pytest.mark.parametrize("linetype", [0, 1])
def test_bilinear_goldens_through_the_c_abi(gpu_lib, linetype):
    from mpassit_amd import regrid as R
    cases = GOLD["bilinear"]
    n = len(cases)
    cells = np.array([v for c in cases for v in c["tri"]])                 # cells 3v .. 3v+2 share vertex v
    latc, lonc = _latlon_rad(cells)
    cen = np.array([np.mean(c["tri"], axis=0) for c in cases])
    latv, lonv = _latlon_rad(cen / np.linalg.norm(cen, axis=1)[:, None])
    voc = np.zeros((3 * n, 3), np.int32)
    voc[:, 0] = np.repeat(np.arange(1, n + 1), 3)
    mesh = R.Mesh(latc, np.where(lonc < 0, lonc + 2 * np.pi, lonc), latv, np.where(lonv < 0, lonv + 2 * np.pi, lonv), voc)
    latp, lonp = _latlon_rad(np.array([c["p"] for c in cases]))
    lat, lon = np.degrees(latp)[None, :], np.degrees(lonp)[None, :]          # a 1 x n "grid" of the target points
    pad = lambda a, ny, nx: np.resize(a, (ny, nx))                           # the staggered companions are not used by this Store
    grid = R.Grid(lon, lat, pad(lon, 2, n + 1), pad(lat, 2, n + 1), pad(lon, 1, n + 1), pad(lat, 1, n + 1), pad(lon, 2, n), pad(lat, 2, n))
    gpu_lib.tune("bilinear_linetype", linetype)
    try:
        rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    finally:
        gpu_lib.tune("bilinear_linetype", 0)
    idx, w = rh.weights()
    checked, worst = 0, 0.0
    for v, c in enumerate(cases):
        if sorted(idx[v].tolist()) != [3 * v, 3 * v + 1, 3 * v + 2]:
            continue                 # the point also lies in another, lower-numbered random triangle: that one wins, by design
        got = np.empty(3)
        got[idx[v] - 3 * v] = w[v]
        worst = max(worst, np.abs(got - np.array(c["w_normal" if linetype else "w"])).max())
        checked += 1
    print("bilinear (linetype %d) through the C-ABI: %d triangles checked, worst |dw| %.2e" % (linetype, checked, worst))
    assert checked >= n - 4 and worst < 5e-12, (checked, worst)
    rh.release()
    mesh.destroy()
    grid.destroy()


@pytest.mark.gpu
def test_conservative_goldens_through_the_c_abi(gpu_lib):
    from mpassit_amd import regrid as R
    worst = {}
    for c in GOLD["conserve"]:
        src = np.array(c["src"])
        nv = src.shape[0]
        latv, lonv = _latlon_rad(src)
        cen = src.mean(axis=0)
        latc, lonc = _latlon_rad((cen / np.linalg.norm(cen))[None, :])
        voc = np.zeros((1, 8), np.int32)
        voc[0, :nv] = np.arange(1, nv + 1)
        wrap = lambda a: np.where(a < 0, a + 2 * np.pi, a)
        mesh = R.Mesh(latc, wrap(lonc), latv, wrap(lonv), voc)
        latq, lonq = _latlon_rad(np.array(c["corners"]))
        lat_c, lon_c = np.degrees(latq).reshape(5, 5), np.degrees(lonq).reshape(5, 5)
        mid = lambda a: 0.25 * (a[:-1, :-1] + a[1:, :-1] + a[:-1, 1:] + a[1:, 1:])     # centres / edges: not used by this Store
        grid = R.Grid(mid(lon_c), mid(lat_c), lon_c, lat_c, lon_c[:-1, :], lat_c[:-1, :], lon_c[:, :-1], lat_c[:, :-1])
        rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_CONSERVE)
        rp, col, val = rh.csr()
        got = np.zeros(16)
        for p in range(16):
            if rp[p + 1] > rp[p]:
                got[p] = val[rp[p]]
        want = np.array(c["w"])
        big = want > 1e-9
        key = (c["scale_km"], c["ratio"])
        worst[key] = max(worst.get(key, 0.0), np.abs(got - want)[big].max())
        rh.release()
        mesh.destroy()
        grid.destroy()
    print("conservative through the C-ABI, worst |dw| per (cell km, dst / src size):", {k: float("%.2e" % v) for k, v in worst.items()})
    for (scale_km, ratio), e in worst.items():
        h = scale_km * ratio / 6370.0
        assert e < 256 * 2.2e-16 / h, (scale_km, ratio, e, worst)      # 1.2e-10 at 3 km (input roundings included); measured below 1e-11


@pytest.mark.gpu
def test_quad_goldens_through_the_c_abi(gpu_lib):
    """CENTER -> EDGE1 destaggering weights of a 2 x 2 CENTER grid whose middle U points are the goldens' points."""
    from mpassit_amd import regrid as R
    worst = 0.0
    for c in GOLD["quad"]:
        latc, lonc = _latlon_rad(np.array(c["centres"]))
        lat, lon = np.degrees(latc).reshape(2, 2), np.degrees(lonc).reshape(2, 2)
        latu, lonu = _latlon_rad(np.array(c["pts"]))
        lat_u, lon_u = np.degrees(latu).reshape(2, 3), np.degrees(lonu).reshape(2, 3)
        pad = lambda a, ny, nx: np.resize(a, (ny, nx))
        grid = R.Grid(lon, lat, pad(lon, 3, 3), pad(lat, 3, 3), lon_u, lat_u, pad(lon, 3, 2), pad(lat, 3, 2))
        rs = R.regrid_store_grid(grid, R.STAGGERLOC_EDGE1)
        idx, w = rs.weights()
        for p in (1, 4):
            assert sorted(idx[p].tolist()) == [0, 1, 2, 3], (c["scale_km"], idx[p])
            got = np.empty(4)
            got[idx[p]] = w[p]
            worst = max(worst, np.abs(got - np.array(c["w"][p])).max())
        rs.release()
        grid.destroy()
    print("quad bilinear through the C-ABI: worst |dw| %.2e" % worst)
    assert worst < 5e-12


@pytest.mark.parametrize("script,fixture", [("make_projection_goldens.py", "projection_wide.json"), ("make_weight_goldens.py", "weights_hp.json"),
                                            ("make_store_goldens.py", "store_hp.json")])
def test_golden_generators_reproduce_the_committed_fixtures(tmp_path, script, fixture):
    """The fixtures are what their committed generators write, byte for byte (mpmath, fixed seeds)."""
    import subprocess
    import sys
    gold = os.path.join(os.path.dirname(__file__), "golden")
    out = str(tmp_path / fixture)
    r = subprocess.run([sys.executable, os.path.join(gold, script), out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert open(out, "rb").read() == open(os.path.join(gold, fixture), "rb").read()
