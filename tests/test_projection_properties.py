"""The target-grid projection against an INDEPENDENT derivation (Snyder, Map Projections -- A Working Manual, eqs 15-1..15-5
for the spherical Lambert conformal conic; plain arithmetic for the lat-lon grid) on thousands of sampled points, plus
round trips and the relations between the four staggers.  The nine compiled-reference numbers of
tests/golden/projection_lc.json pin one Lambert grid bit-wise; a wider golden set cannot be generated here (the reference's
projection modules need utils.F90 -> MPI + NetCDF), so the rest of the parameter space -- secant cones,
southern hemisphere, every stagger, lat-lon regional and global -- is pinned by construction instead.  CPU only."""
import numpy as np
import pytest

R_EARTH = 6370000.0          # constants_module.F90: EARTH_RADIUS_M


def snyder_lcc_xy(lat, lon, truelat1, truelat2, stdlon, lat0):
    """Projected (x, y) in metres relative to the point (lat0, stdlon); spherical LCC, Snyder 15-1..15-5."""
    p1, p2, p0, ph, lam = map(np.deg2rad, (truelat1, truelat2, lat0, lat, ((lon - stdlon + 180.0) % 360.0) - 180.0))
    sgn = 1.0 if truelat1 >= 0 else -1.0
    p1, p2, p0, ph = sgn * p1, sgn * p2, sgn * p0, sgn * ph            # southern hemisphere: mirror
    if abs(truelat1 - truelat2) > 1e-9:
        n = np.log(np.cos(p1) / np.cos(p2)) / np.log(np.tan(np.pi / 4 + p2 / 2) / np.tan(np.pi / 4 + p1 / 2))
    else:
        n = np.sin(p1)
    F = np.cos(p1) * np.tan(np.pi / 4 + p1 / 2) ** n / n
    rho = R_EARTH * F / np.tan(np.pi / 4 + ph / 2) ** n
    rho0 = R_EARTH * F / np.tan(np.pi / 4 + p0 / 2) ** n
    th = n * lam * sgn
    return rho * np.sin(th) * sgn, sgn * (rho0 - rho * np.cos(th)), n


LAMBERTS = [dict(truelat1=38.5, truelat2=38.5, stdlon=-97.5, lat1=38.5, lon1=-97.5, knowni=900.5, knownj=530.5, dx=3000.0),      # README grid
            dict(truelat1=30.0, truelat2=60.0, stdlon=-98.0, lat1=34.0, lon1=-101.0, knowni=150.0, knownj=120.0, dx=12000.0),    # secant, off-centre
            dict(truelat1=-30.0, truelat2=-60.0, stdlon=25.0, lat1=-40.0, lon1=20.0, knowni=80.5, knownj=60.5, dx=9000.0),       # southern hemisphere
            dict(truelat1=60.0, truelat2=30.0, stdlon=10.0, lat1=50.0, lon1=8.0, knowni=200.0, knownj=180.0, dx=4000.0)]         # true latitudes swapped


@pytest.mark.parametrize("k", range(len(LAMBERTS)))
def test_lambert_against_snyder_and_round_trips(oracle, k):
    from mpassit_amd import target_grid as tg
    P = LAMBERTS[k]
    po = oracle.Proj.lambert(P["truelat1"], P["truelat2"], P["stdlon"], P["lat1"], P["lon1"], P["knowni"], P["knownj"], P["dx"])
    ph = tg.Proj.lambert(P["truelat1"], P["truelat2"], P["stdlon"], P["lat1"], P["lon1"], P["knowni"], P["knownj"], P["dx"])
    rng = np.random.default_rng(100 + k)
    xi = rng.uniform(1.0, 2.0 * P["knowni"], 1500)
    yj = rng.uniform(1.0, 2.0 * P["knownj"], 1500)
    lat_h, lon_h = ph.ij_to_latlon(xi, yj)                               # numpy mirror, vectorised
    x0, y0, n = snyder_lcc_xy(P["lat1"], P["lon1"], P["truelat1"], P["truelat2"], P["stdlon"], P["lat1"])
    for q in range(0, 1500, 7):                                           # the oracle's scalar path on a subsample
        la, lo = po.xytoll(xi[q], yj[q], oracle.M)
        assert abs(la - lat_h[q]) < 1e-11 and abs(((lo - lon_h[q] + 180.0) % 360.0) - 180.0) < 1e-11
        i2, j2 = po.latlon_to_ij(la, lo)                                  # round trip
        assert abs(i2 - xi[q]) < 1e-7 and abs(j2 - yj[q]) < 1e-7
    # independent derivation: the projected plane coordinates of the returned lat/lon are the grid offsets times dx
    x, y, _ = snyder_lcc_xy(lat_h, lon_h, P["truelat1"], P["truelat2"], P["stdlon"], P["lat1"])
    assert np.abs((x - x0) - (xi - P["knowni"]) * P["dx"]).max() < 1e-5   # metres
    assert np.abs((y - y0) - (yj - P["knownj"]) * P["dx"]).max() < 1e-5
    assert abs(po.fields()["cone"] - n) < 1e-13
    # stagger relations (llxy_module.F90:166-216): U(i, j) = M(i - 1/2, j), V(i, j) = M(i, j - 1/2), CORNER(i, j) = M(i - 1/2, j - 1/2)
    for q in range(0, 1500, 97):
        i, j = float(int(xi[q])), float(int(yj[q]))
        for st, di, dj in ((oracle.U, -0.5, 0.0), (oracle.V, 0.0, -0.5), (oracle.CORNER, -0.5, -0.5)):
            a, b = po.xytoll(i, j, st), po.xytoll(i + di, j + dj, oracle.M)
            assert abs(a[0] - b[0]) < 1e-12 and abs(a[1] - b[1]) < 1e-12


@pytest.mark.parametrize("glob", [False, True])
def test_latlon_grid_is_plain_arithmetic(oracle, glob):
    from mpassit_amd import target_grid as tg
    if glob:
        g = tg.define_target_grid_params("lat-lon", 73, 37, stand_lon=-180.0, is_regional=False)
        dlat = dlon = 5.0
    else:
        g = tg.define_target_grid_params("lat-lon", 41, 31, dx=0.25, dy=0.25, ref_lat=30.0, ref_lon=-110.0, ref_x=1.0, ref_y=1.0, stand_lon=-110.0)
        dlat = dlon = 0.25
    assert np.abs(np.diff(g.lat[:, 0]) - dlat).max() < 1e-12 and np.abs(np.diff(g.lon[0, :]) - dlon).max() < 1e-12
    assert np.abs(g.lat_v[:-1] - (g.lat - dlat / 2)).max() < 1e-12 and np.abs(g.lon_u[:, :-1] - (g.lon - dlon / 2)).max() < 1e-12
    assert np.abs(g.lat_c[:-1, :-1] - (g.lat - dlat / 2)).max() < 1e-12 and np.abs(g.lon_c[:-1, :-1] - (g.lon - dlon / 2)).max() < 1e-12
    if glob:
        assert g.lat_c[0, 0] == -90.0 and g.lat_c[-1, 0] == 90.0 and abs(((g.lon_c[0, -1] - g.lon_c[0, 0]) + 180.0) % 360.0 - 180.0) < 1e-12   # the duplicate column, one period later
