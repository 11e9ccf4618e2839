#!/usr/bin/env python3
"""Generator of tests/golden/projection_wide.json -- an INDEPENDENT, reproducible pin of the target-grid projections.

The reference's own projection modules cannot be compiled here (module_map_utils.F90 needs utils_mod, which needs the MPI
and NetCDF Fortran modules: DESIGN.md s2), so no vectors can come from it.  What this script pins instead is the
MATHEMATICAL DEFINITION the reference implements: the spherical Lambert conformal (one or two standard parallels), polar
stereographic, Mercator and equidistant cylindrical (lat-lon) projections of Snyder, "Map Projections -- A Working Manual"
(USGS PP 1395, 1987; eqs. 15-1..15-11, 21-1..21-15 with the scale true at a standard parallel, 7-1..7-5 scaled by the
standard parallel), parametrised the way map_set is called (llxy_module.F90:60-132): earth radius 6 370 000 m, grid length dx,
a known point (lat1, lon1) at index (knowni, knownj), standard longitude, true latitudes.  Nothing of the code under test is
imported: the formulas are written out below and evaluated with mpmath at 40 digits, then rounded once to float64.

Every record is one point: projection set-up, stagger (M / U / V / CORNER, xytoll: llxy_module.F90:166-216), 1-based index
(i, j) of get_lat_lon_fields (model_grid.F90:2188-2219) -> latitude, longitude (degrees, longitude in [-180, 180]) and the
map scale factor (get_map_factor's quantity: grid length / ground length).  tests/test_projection_goldens.py holds the oracle,
the numpy mirror and (on the GPU box) the device kernel against them.

    python tests/golden/make_projection_goldens.py        # rewrites tests/golden/projection_wide.json (deterministic)
"""
import json
import os

import mpmath as mp

mp.mp.dps = 40
R = mp.mpf(6370000)
D2R = mp.pi / 180

SETUPS = [
    dict(name="lc_readme", kind="lambert", nx=1801, ny=1061, dx=3000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5),
    dict(name="lc_two_truelats_south", kind="lambert", nx=301, ny=201, dx=12000.0, ref_lat=-35.0, ref_lon=140.0, truelat1=-30.0, truelat2=-60.0,
         stand_lon=135.0, ref_x=100.0, ref_y=80.0),
    dict(name="lc_two_truelats_dateline", kind="lambert", nx=401, ny=301, dx=15000.0, ref_lat=52.0, ref_lon=178.0, truelat1=45.0, truelat2=60.0,
         stand_lon=-175.0),
    dict(name="ps_north_pole_inside", kind="polar", nx=401, ny=351, dx=15000.0, ref_lat=78.0, ref_lon=-40.0, truelat1=70.0, stand_lon=-45.0),
    dict(name="ps_south", kind="polar", nx=201, ny=221, dx=25000.0, ref_lat=-75.0, ref_lon=100.0, truelat1=-71.0, stand_lon=90.0, ref_x=60.0,
         ref_y=80.0),
    dict(name="merc_dateline", kind="mercator", nx=501, ny=301, dx=12000.0, ref_lat=8.0, ref_lon=150.0, truelat1=15.0, stand_lon=150.0),
    dict(name="merc_south", kind="mercator", nx=201, ny=151, dx=20000.0, ref_lat=-20.0, ref_lon=-70.0, truelat1=-30.0, stand_lon=-70.0, ref_x=1.0,
         ref_y=1.0),
    dict(name="latlon_regional", kind="lat-lon", nx=201, ny=201, dx=0.1, dy=0.1, ref_lat=30.0, ref_lon=-110.0, ref_x=1.0, ref_y=1.0,
         stand_lon=-110.0),
    dict(name="latlon_global", kind="lat-lon", nx=361, ny=181, stand_lon=0.0, is_regional=False),
]
STAGGERS = {"M": (0.0, 0.0), "U": (0.5, 0.0), "V": (0.0, 0.5), "CORNER": (0.5, 0.5)}   # what xytoll subtracts from (x, y)


def wrap180(x):
    x = mp.mpf(x)
    while x > 180:
        x -= 360
    while x < -180:
        x += 360
    return x


def known_point(s):
    """program_setup.F90:160-164,242-249: the mass grid is (nx-1) x (ny-1); the known point defaults to its centre."""
    ni, nj = s["nx"] - 1, s["ny"] - 1
    return mp.mpf(s.get("ref_x", (ni + 1) / 2.0)), mp.mpf(s.get("ref_y", (nj + 1) / 2.0))


def psi(phi):                      # isometric latitude on the sphere
    return mp.log(mp.tan(mp.pi / 4 + phi / 2))


def lambert(s, i, j):
    p1, p2 = mp.mpf(s["truelat1"]) * D2R, mp.mpf(s["truelat2"]) * D2R
    lam0 = mp.mpf(s["stand_lon"]) * D2R
    if abs(s["truelat1"] - s["truelat2"]) > 0.1:           # Snyder 15-3
        n = mp.log(mp.cos(p1) / mp.cos(p2)) / mp.log(mp.tan(mp.pi / 4 + p2 / 2) / mp.tan(mp.pi / 4 + p1 / 2))
    else:
        n = mp.sin(p1)
    F = mp.cos(p1) * mp.tan(mp.pi / 4 + p1 / 2) ** n / n   # 15-2
    rho = lambda phi: R * F / mp.tan(mp.pi / 4 + phi / 2) ** n   # 15-1
    dx = mp.mpf(s["dx"])
    ki, kj = known_point(s)
    phik, lamk = mp.mpf(s["ref_lat"]) * D2R, mp.mpf(s["ref_lon"]) * D2R
    dl = wrap180((lamk - lam0) / D2R) * D2R
    # plane coordinates with the cone apex at the origin: x = rho sin(theta), y = -rho cos(theta)
    xk, yk = rho(phik) * mp.sin(n * dl), -rho(phik) * mp.cos(n * dl)
    x, y = xk + (i - ki) * dx, yk + (j - kj) * dx
    r = mp.sqrt(x * x + y * y)
    if r == 0:
        return mp.sign(n) * 90, wrap180(s["stand_lon"]), None
    r_signed = mp.sign(n) * r                              # 15-10
    theta = mp.atan2(mp.sign(n) * x, -mp.sign(n) * y)      # 15-11 (with the sign convention of a southern cone)
    phi = 2 * mp.atan((R * F / r_signed) ** (1 / n)) - mp.pi / 2   # 15-5
    lam = lam0 + theta / n
    m = mp.cos(p1) / mp.cos(phi) * (mp.tan(mp.pi / 4 + p1 / 2) / mp.tan(mp.pi / 4 + phi / 2)) ** n   # k of 15-4, = 1 at phi_1
    return phi / D2R, wrap180(lam / D2R), m


def polar(s, i, j):
    h = -1 if s["truelat1"] < 0 else 1
    p1 = abs(mp.mpf(s["truelat1"])) * D2R
    lam0 = mp.mpf(s["stand_lon"]) * D2R
    rho = lambda phi: R * (1 + mp.sin(p1)) * mp.tan(mp.pi / 4 - h * phi / 2)     # 21-33 with k0 = (1 + sin phi_c) / 2
    dx = mp.mpf(s["dx"])
    ki, kj = known_point(s)
    phik, lamk = mp.mpf(s["ref_lat"]) * D2R, mp.mpf(s["ref_lon"]) * D2R
    xk, yk = rho(phik) * mp.sin(lamk - lam0), -h * rho(phik) * mp.cos(lamk - lam0)   # 21-5/6 (north), 21-9/10 (south)
    x, y = xk + (i - ki) * dx, yk + (j - kj) * dx
    r = mp.sqrt(x * x + y * y)
    if r == 0:
        return mp.mpf(h * 90), None, None
    phi = h * (mp.pi / 2 - 2 * mp.atan(r / (R * (1 + mp.sin(p1)))))
    lam = lam0 + mp.atan2(x, -h * y)
    m = (1 + mp.sin(p1)) / (1 + mp.sin(h * phi))
    return phi / D2R, wrap180(lam / D2R), m


def mercator(s, i, j):
    p1 = mp.mpf(s["truelat1"]) * D2R
    dx = mp.mpf(s["dx"])
    ki, kj = known_point(s)
    phik, lamk = mp.mpf(s["ref_lat"]) * D2R, mp.mpf(s["ref_lon"]) * D2R
    a = R * mp.cos(p1)                                     # 7-1 / 7-2 on a cylinder secant at phi_1
    lam = lamk + (i - ki) * dx / a
    phi = 2 * mp.atan(mp.exp(psi(phik) + (j - kj) * dx / a)) - mp.pi / 2   # 7-4
    return phi / D2R, wrap180(lam / D2R), mp.cos(p1) / mp.cos(phi)


def latlon(s, i, j):
    ni, nj = s["nx"] - 1, s["ny"] - 1
    if "dx" in s:
        dlon, dlat = mp.mpf(s["dx"]), mp.mpf(s["dy"])
        ki, kj = known_point(s)
        lat1, lon1 = mp.mpf(s["ref_lat"]), mp.mpf(s["ref_lon"])
    else:                                                  # global: program_setup.F90:197-210
        dlon, dlat = mp.mpf(360) / ni, mp.mpf(180) / nj
        ki = kj = mp.mpf(1)
        lon1, lat1 = mp.mpf(s["stand_lon"]) + dlon / 2, -90 + dlat / 2
    return lat1 + (j - kj) * dlat, lon1 + (i - ki) * dlon, None   # the longitude of a lat-lon grid is not wrapped (ijll_latlon)


FUN = {"lambert": lambert, "polar": polar, "mercator": mercator, "lat-lon": latlon}


def main():
    import random
    rng = random.Random(20261004)
    out = dict(earth_radius_m=6370000.0, note="generated by tests/golden/make_projection_goldens.py (mpmath, 40 digits); see its docstring",
               setups=[])
    for s in SETUPS:
        ni, nj = s["nx"] - 1, s["ny"] - 1
        pts = []
        for st, (ox, oy) in STAGGERS.items():
            n_i = ni + (1 if st in ("U", "CORNER") else 0)
            n_j = nj + (1 if st in ("V", "CORNER") else 0)
            picks = {(1, 1), (n_i, 1), (1, n_j), (n_i, n_j), ((n_i + 1) // 2, (n_j + 1) // 2)}
            while len(picks) < 130:
                picks.add((rng.randint(1, n_i), rng.randint(1, n_j)))
            for (i, j) in sorted(picks):
                lat, lon, m = FUN[s["kind"]](s, mp.mpf(i) - ox, mp.mpf(j) - oy)
                pts.append([st, i, j, float(lat), None if lon is None else float(lon), None if m is None else float(m)])
        out["setups"].append(dict(params={k: v for k, v in s.items()}, points=pts))
    import sys
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "projection_wide.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print("wrote %s: %d set-ups, %d points" % (path, len(out["setups"]), sum(len(x["points"]) for x in out["setups"])))


if __name__ == "__main__":
    main()
