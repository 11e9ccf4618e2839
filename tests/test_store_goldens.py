"""Whole RegridStore results of the oracle against tests/golden/store_hp.json -- three small mesh / grid pairs solved by brute force at
50 digits (tests/golden/make_store_goldens.py: every target point against every dual triangle, every cell pair clipped, Girard areas;
no search structure, no float64 geometry).  What the formula goldens (weights_hp.json) cannot see is pinned here: which triangle holds
a point, the rim and the unmapped points of a regional mesh, the date line, a pole inside a cell, which pairs overlap, how a
conservative row is normalised.  tests/test_store_goldens_gpu.py asks the same of the library through the C-ABI."""
import json
import os

import numpy as np
import pytest

from conftest import mesh_xyz

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "store_hp.json")))
# The float64 latitude / longitude -> unit vector step inside the library (and the oracle) moves a point by ~1e-16 of the radius, i.e. by
# 1e-16 / h of a cell of angular size h: weights can agree with the 50-digit answer to a few times that and no better.  Measured: 5e-14 at
# 120 km, 2e-12 at 3 km (bilinear), 2e-14 / 1e-12 (conservative).  north_star's bar is 1e-6.
def tol_for(case):
    return max(2e-13, 4e-15 / case.h)


SLIVER = 1e-12         # entries below this may exist on one side only (both implementations drop overlaps < 1e-14 of the cell)


class Case:
    def __init__(self, c):
        from mpassit_amd import synth
        self.name, self.nx, self.ny = c["name"], c["nx"], c["ny"]
        self.mesh = synth.MpasMesh(np.array(c["latCell"]), np.array(c["lonCell"]), np.array(c["latVertex"]), np.array(c["lonVertex"]),
                                   np.array(c["verticesOnCell"], np.int32))
        sh, shc = (self.ny, self.nx), (self.ny + 1, self.nx + 1)
        self.lon, self.lat = np.array(c["lon"]).reshape(sh), np.array(c["lat"]).reshape(sh)
        self.lon_c, self.lat_c = np.array(c["lon_c"]).reshape(shc), np.array(c["lat_c"]).reshape(shc)
        self.lon_u, self.lat_u = np.array(c["lon_u"]).reshape(self.ny, self.nx + 1), np.array(c["lat_u"]).reshape(self.ny, self.nx + 1)
        self.lon_v, self.lat_v = np.array(c["lon_v"]).reshape(self.ny + 1, self.nx), np.array(c["lat_v"]).reshape(self.ny + 1, self.nx)
        self.expect = c["expect"]
        a, b = np.radians([self.lon[0, 0], self.lat[0, 0]]), np.radians([self.lon[0, 1], self.lat[0, 1]])
        xyz = lambda q: np.array([np.cos(q[1]) * np.cos(q[0]), np.cos(q[1]) * np.sin(q[0]), np.sin(q[1])])   # noqa: E731
        self.h = float(np.linalg.norm(xyz(a) - xyz(b)))                  # grid spacing, radians


def cases():
    return [Case(c) for c in GOLD["cases"]]


def check_bilinear(case, idx, w):
    """idx / w [P][3] (-1 = unmapped) against the golden: same mapped mask, same three cells, weights within BIL_TOL."""
    worst, n = 0.0, 0
    for p, e in enumerate(case.expect["bilinear"]):
        if e is None:
            assert idx[p, 0] < 0, "%s: point %d lies in no dual triangle, yet is mapped to %s" % (case.name, p, idx[p])
            continue
        if "tie" in e:
            continue
        assert idx[p, 0] >= 0, "%s: point %d lies in triangle %s, yet is unmapped" % (case.name, p, e["col"])
        order = np.argsort(idx[p])
        assert list(idx[p][order]) == e["col"], "%s: point %d: cells %s, expected %s" % (case.name, p, idx[p][order], e["col"])
        worst = max(worst, np.abs(w[p][order] - np.array(e["w"])).max())
        n += 1
    assert n > 50 and worst < tol_for(case), (case.name, n, worst)
    return worst


def check_nearest(case, idx):
    for p, e in enumerate(case.expect["nearest"]):
        if "tie" not in e:
            assert int(idx[p]) == e["col"], "%s: point %d: nearest cell %d, expected %d" % (case.name, p, idx[p], e["col"])


def check_conserve(case, rowptr, col, val):
    got = {(int(r), int(c)): float(v) for r in range(rowptr.size - 1) for c, v in zip(col[rowptr[r]:rowptr[r + 1]], val[rowptr[r]:rowptr[r + 1]])}
    want = {(r, c): v for r, c, v in case.expect["conserve"]}
    worst = 0.0
    for k, v in want.items():
        if k in got:
            worst = max(worst, abs(got[k] - v))
        else:
            assert v < SLIVER, "%s: overlap %s = %.3e is missing" % (case.name, k, v)
    for k, v in got.items():
        assert k in want or v < SLIVER, "%s: entry %s = %.3e overlaps nothing" % (case.name, k, v)
    assert len(want) > 100 and worst < tol_for(case), (case.name, worst)
    # rows of fully covered destination cells sum to 1 in the golden itself (Girard) -- and so must ours
    rows = {}
    for (r, _), v in want.items():
        rows[r] = rows.get(r, 0.0) + v
    full = [r for r, s in rows.items() if abs(s - 1.0) < 1e-9]
    assert len(full) >= 10
    for r in full:
        assert abs(val[rowptr[r]:rowptr[r + 1]].sum() - 1.0) < 1e-11
    return worst


@pytest.mark.parametrize("case", cases(), ids=lambda c: c.name)
def test_oracle_whole_stores_equal_the_brute_force_goldens(oracle, case):
    o, m = oracle, case.mesh
    cxyz, vxyz = mesh_xyz(o, m)
    tri, _ = o.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    pxyz = o.lonlat_deg_to_xyz(case.lon, case.lat)
    idx, w = o.bilinear_weights(cxyz, tri, pxyz)
    wb = check_bilinear(case, idx, w)
    check_nearest(case, o.nearest(cxyz, pxyz))
    kxyz = o.lonlat_deg_to_xyz(case.lon_c, case.lat_c)
    rowptr, col, val = o.conserve(m.verticesOnCell, vxyz, case.nx, case.ny, kxyz)[:3]
    wc = check_conserve(case, rowptr, col, val)
    print("%s: oracle vs 50-digit brute force: bilinear %.1e, conservative %.1e" % (case.name, wb, wc))


def test_the_goldens_cover_what_they_claim():
    """rim + unmapped points in the regional case, the date line in the second, the pole inside the third grid."""
    c1, c2, c3 = cases()[:3]
    assert sum(e is None for e in c1.expect["bilinear"]) > 20 and sum(e is not None for e in c1.expect["bilinear"]) > 100
    lon = np.mod(c2.lon, 360.0)
    assert lon.min() < 180.0 < lon.max()
    assert c3.lat.max() > 85.0 and np.ptp(c3.lon) > 300.0
    assert {4, 5, 6, 7} <= set(((np.array(c2.mesh.verticesOnCell) > 0).sum(1)).tolist())


def check_grid_to_grid(case, stagger_key, idx, w):
    """idx / w [P][4] (-1 = unmapped) against the golden's sparse weight vectors; returns (worst difference, points that are mapped on one
    side only, with their distance-from-the-hull diagnostics left to the caller)."""
    exp = case.expect[stagger_key]
    worst, one_sided = 0.0, []
    for p, e in enumerate(exp):
        got = {}
        for c, v in zip(idx[p], w[p]):
            if c >= 0 and abs(v) > 1e-13:
                got[int(c)] = got.get(int(c), 0.0) + float(v)
        if e is not None and "hull" in e:          # on the hull of the CENTER points to rounding: mapped or not is a tolerance's business
            continue
        if e is None or not got:
            if (e is None) != (not got):
                one_sided.append(p)
            continue
        want = dict(zip(e["col"], e["w"]))
        d = max(abs(want.get(k, 0.0) - got.get(k, 0.0)) for k in set(want) | set(got))
        if e.get("tie"):       # within 1e-9 of the border of two quads: their vectors differ by that distance at most, and either quad is right
            assert d < 3e-9, (case.name, stagger_key, p, d)
        else:
            worst = max(worst, d)
    return worst, one_sided


@pytest.mark.parametrize("case", cases(), ids=lambda c: c.name)
def test_oracle_destaggering_stores_equal_the_brute_force_goldens(oracle, case):
    o = oracle
    cxyz = o.lonlat_deg_to_xyz(case.lon, case.lat)
    for key, stag, lon_e, lat_e in (("edge1", 1, case.lon_u, case.lat_u), ("edge2", 2, case.lon_v, case.lat_v)):
        idx, w = o.grid_bilinear(case.nx, case.ny, cxyz, stag, o.lonlat_deg_to_xyz(lon_e, lat_e))
        worst, one_sided = check_grid_to_grid(case, key, idx, w)
        print("%s %s: oracle vs 50-digit brute force %.1e, mapped on one side only: %d" % (case.name, key, worst, len(one_sided)))
        assert worst < 10 * tol_for(case), (case.name, key, worst)
        assert not one_sided, (case.name, key, one_sided[:10])
