"""The target-grid projections against tests/golden/projection_wide.json: 4 680 points (nine set-ups x four staggers) computed
by tests/golden/make_projection_goldens.py from the projections' mathematical definitions (Snyder's spherical formulas,
mpmath at 40 digits) -- independent of the reference's code (which cannot be compiled here), of the oracle and of
the product.  Oracle (C restatement of module_map_utils.F90) and numpy mirror here; the device kernel in the GPU half.
Tolerances: 2e-11 degrees (float64 round-off through ~20 operations incl. pow / atan near a pole), map factors 1e-12."""
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "projection_wide.json")))
DEG_TOL = 2e-11
STAG = {"M": 1, "U": 2, "V": 3, "CORNER": 4}          # oracle's xytoll codes
STAG_T = {"M": 1, "U": 2, "V": 3, "CORNER": 6}        # target_grid's (misc_definitions_module.F90:29)


def _dlon(a, b):
    d = np.abs(np.asarray(a) - np.asarray(b)) % 360.0
    return np.minimum(d, 360.0 - d)


def _params(s):
    p = dict(s["params"])
    kind = p.pop("kind")
    p.pop("name")
    return kind, p


def _mirror_grid(s):
    from mpassit_amd import target_grid as T
    kind, p = _params(s)
    return T.define_target_grid_params(kind, p.pop("nx"), p.pop("ny"), arrays=False, **p)


@pytest.mark.parametrize("k", range(len(GOLD["setups"])))
def test_numpy_mirror_and_oracle_against_independent_goldens(oracle, k):
    from mpassit_amd import target_grid as T
    s = GOLD["setups"][k]
    g = _mirror_grid(s)
    pr = g.proj
    if pr.code == T.PROJ_LC:
        po = oracle.Proj.lambert(pr.truelat1, pr.truelat2, pr.stdlon, pr.lat1, pr.lon1, pr.knowni, pr.knownj, pr.dx)
    elif pr.code == T.PROJ_PS:
        po = oracle.Proj.polar(pr.truelat1, pr.stdlon, pr.lat1, pr.lon1, pr.knowni, pr.knownj, pr.dx)
    elif pr.code == T.PROJ_MERC:
        po = oracle.Proj.mercator(pr.truelat1, pr.lat1, pr.lon1, pr.knowni, pr.knownj, pr.dx)
    else:
        po = oracle.Proj.latlon(pr.lat1, pr.lon1, pr.knowni, pr.knownj, pr.latinc, pr.loninc)
    n = 0
    for st, i, j, lat, lon, mf in s["points"]:
        la_m, lo_m = pr.xytoll(float(i), float(j), STAG_T[st])
        la_o, lo_o = po.xytoll(float(i), float(j), STAG[st])
        assert abs(float(la_m) - lat) < DEG_TOL and abs(la_o - lat) < DEG_TOL, (s["params"]["name"], st, i, j, float(la_m), la_o, lat)
        if lon is not None:
            assert _dlon(float(lo_m), lon) < DEG_TOL and _dlon(lo_o, lon) < DEG_TOL, (s["params"]["name"], st, i, j, float(lo_m), lo_o, lon)
        if mf is not None:
            assert abs(float(T.get_map_factor(pr, lat)) / mf - 1) < 1e-12 and abs(float(po.map_factor(lat)) / mf - 1) < 1e-12
        if pr.code != T.PROJ_LATLON and lon is not None:       # and back: latlon_to_ij of the golden coordinates
            off = {"M": (0.0, 0.0), "U": (0.5, 0.0), "V": (0.0, 0.5), "CORNER": (0.5, 0.5)}[st]
            i2, j2 = pr.latlon_to_ij(lat, lon)
            assert abs(float(i2) - (i - off[0])) < 1e-6 and abs(float(j2) - (j - off[1])) < 1e-6
        n += 1
    assert n == 520


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(GOLD["setups"])))
def test_device_grid_against_independent_goldens(gpu_lib, k):
    from mpassit_amd import regrid as R
    s = GOLD["setups"][k]
    g = _mirror_grid(s)
    grid = R.Grid.from_proj(g)
    arrs = {st: grid.coords(loc) for st, loc in (("M", R.STAGGERLOC_CENTER), ("U", R.STAGGERLOC_EDGE1), ("V", R.STAGGERLOC_EDGE2),
                                                 ("CORNER", R.STAGGERLOC_CORNER))}
    mfs = {st: grid.mapfac(loc) for st, loc in (("M", R.STAGGERLOC_CENTER), ("U", R.STAGGERLOC_EDGE1), ("V", R.STAGGERLOC_EDGE2))}
    for st, i, j, lat, lon, mf in s["points"]:
        lon_d, lat_d = arrs[st]
        assert abs(lat_d[j - 1, i - 1] - lat) < DEG_TOL, (s["params"]["name"], st, i, j)
        if lon is not None:
            assert _dlon(lon_d[j - 1, i - 1], lon) < DEG_TOL, (s["params"]["name"], st, i, j)
        if mf is not None and st in mfs:
            assert abs(mfs[st][j - 1, i - 1] / mf - 1) < 1e-12
    grid.destroy()
