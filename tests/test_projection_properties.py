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


def _haversine_m(lat1, lon1, lat2, lon2, R=6370000.0):
    p1, p2 = np.radians(lat1), np.radians(lat2)
    a = np.sin((p2 - p1) / 2) ** 2 + np.cos(p1) * np.cos(p2) * np.sin(np.radians(lon2 - lon1) / 2) ** 2
    return 2 * R * np.arcsin(np.sqrt(a))


POLARS = [dict(truelat1=70.0, stdlon=-45.0, lat1=78.0, lon1=-40.0, knowni=200.5, knownj=175.5, dx=15000.0),
          dict(truelat1=-71.0, stdlon=90.0, lat1=-75.0, lon1=100.0, knowni=60.0, knownj=80.0, dx=25000.0),
          dict(truelat1=90.0, stdlon=0.0, lat1=60.0, lon1=10.0, knowni=1.0, knownj=1.0, dx=50000.0)]


@pytest.mark.parametrize("k", range(len(POLARS)))
def test_polar_stereographic_against_snyder_and_round_trips(oracle, k):
    """set_ps / ijll_ps / llij_ps (module_map_utils.F90:682-822) as the numpy mirror and the oracle restate them, against
    an independent derivation: on a sphere the polar stereographic radius of latitude phi, true at phi_1, is
    rho = R (1 + sin|phi_1|) tan(pi/4 - |phi|/2) (Snyder 1987, eq. 21-33 with k_0 from the standard parallel), the azimuth
    is the longitude itself; plus the known point, round trips, and the map factor as the ratio grid length : ground length."""
    from mpassit_amd import target_grid as T
    c = POLARS[k]
    pm = T.Proj.polar(**c)
    po = oracle.Proj.polar(c["truelat1"], c["stdlon"], c["lat1"], c["lon1"], c["knowni"], c["knownj"], c["dx"])
    rng = np.random.default_rng(70 + k)
    ii, jj = rng.uniform(-50, 450, 1500), rng.uniform(-50, 400, 1500)
    lat, lon = pm.ij_to_latlon(ii, jj)
    for q in range(0, 1500, 50):                                   # mirror == oracle, point by point
        la, lo = po.xytoll(ii[q], jj[q])
        assert abs(la - lat[q]) < 1e-12 and min(abs(lo - lon[q]), 360 - abs(lo - lon[q])) < 1e-12
    la, lo = pm.ij_to_latlon(c["knowni"], c["knownj"])           # the known point
    assert abs(la - c["lat1"]) < 1e-10 and abs(lo - c["lon1"]) < 1e-10
    i2, j2 = pm.latlon_to_ij(lat, lon)                             # round trip
    assert np.abs(i2 - ii).max() < 1e-7 and np.abs(j2 - jj).max() < 1e-7
    io, jo = po.latlon_to_ij(float(lat[3]), float(lon[3]))
    assert abs(io - ii[3]) < 1e-7 and abs(jo - jj[3]) < 1e-7
    hemi = -1.0 if c["truelat1"] < 0 else 1.0                      # Snyder: distance from the pole in grid lengths
    rho = 6370000.0 * (1 + np.sin(np.radians(abs(c["truelat1"])))) * np.tan(np.pi / 4 - np.radians(hemi * lat) / 2) / c["dx"]
    assert np.abs(np.hypot(ii - pm.polei, jj - pm.polej) - rho).max() < 1e-7
    pole_lat, _ = pm.ij_to_latlon(pm.polei, pm.polej)
    assert abs(pole_lat - hemi * 90.0) < 1e-9
    # conformal: one grid length east / north of a point is dx / mapfac on the ground (finite step of 1e-3 grid lengths)
    h = 1e-3
    mf = T.get_map_factor(pm, lat)
    np.testing.assert_allclose(po.map_factor(lat[:40]), mf[:40], rtol=1e-13)
    for di, dj in ((h, 0.0), (0.0, h)):
        la2, lo2 = pm.ij_to_latlon(ii + di, jj + dj)
        ground = _haversine_m(lat, lon, la2, lo2) / h
        far = np.hypot(ii - pm.polei, jj - pm.polej) > 2.0         # away from the pole itself
        np.testing.assert_allclose(ground[far], (c["dx"] / mf)[far], rtol=2e-5)


MERCS = [dict(truelat1=15.0, lat1=8.0, lon1=150.0, knowni=250.5, knownj=150.5, dx=12000.0),
         dict(truelat1=-30.0, lat1=-20.0, lon1=-70.0, knowni=1.0, knownj=1.0, dx=20000.0),
         dict(truelat1=0.0, lat1=0.0, lon1=0.0, knowni=10.0, knownj=20.0, dx=50000.0)]


@pytest.mark.parametrize("k", range(len(MERCS)))
def test_mercator_against_closed_form_and_round_trips(oracle, k):
    """set_merc / ijll_merc / llij_merc (module_map_utils.F90:1293-1362): x = R cos(phi_1) (lambda - lambda_1),
    y = R cos(phi_1) (psi(phi) - psi(phi_1)) with the isometric latitude psi = ln tan(pi/4 + phi/2) (Snyder eq. 7-1, 7-2 scaled
    by the standard parallel); known point, round trips, oracle == mirror, map factor = grid length : ground length."""
    from mpassit_amd import target_grid as T
    c = MERCS[k]
    pm = T.Proj.mercator(**c)
    po = oracle.Proj.mercator(c["truelat1"], c["lat1"], c["lon1"], c["knowni"], c["knownj"], c["dx"])
    rng = np.random.default_rng(80 + k)
    ii, jj = rng.uniform(-100, 600, 1500), rng.uniform(-100, 400, 1500)
    lat, lon = pm.ij_to_latlon(ii, jj)
    assert np.abs(lat).max() < 90.0
    for q in range(0, 1500, 50):
        la, lo = po.xytoll(ii[q], jj[q])
        assert abs(la - lat[q]) < 1e-12 and min(abs(lo - lon[q]), 360 - abs(lo - lon[q])) < 1e-12
    la, lo = pm.ij_to_latlon(c["knowni"], c["knownj"])
    assert abs(la - c["lat1"]) < 1e-10 and abs(lo - c["lon1"]) < 1e-10
    i2, j2 = pm.latlon_to_ij(lat, lon)
    inside = np.abs((ii - c["knowni"]) * pm.dlon) < np.pi           # the longitude wraps once around the globe
    assert np.abs(i2 - ii)[inside].max() < 1e-7 and np.abs(j2 - jj).max() < 1e-7
    Rc = 6370000.0 * np.cos(np.radians(c["truelat1"])) / c["dx"]
    psi = lambda phi: np.log(np.tan(np.pi / 4 + np.radians(phi) / 2))
    assert np.abs((jj - c["knownj"]) - Rc * (psi(lat) - psi(c["lat1"]))).max() < 1e-7
    dl = (lon - c["lon1"] + 540.0) % 360.0 - 180.0
    assert np.abs(((ii - c["knowni"]) - Rc * np.radians(dl))[inside]).max() < 1e-7
    h = 1e-3
    mf = T.get_map_factor(pm, lat)
    np.testing.assert_allclose(po.map_factor(lat[:40]), mf[:40], rtol=1e-13)
    for di, dj in ((h, 0.0), (0.0, h)):
        la2, lo2 = pm.ij_to_latlon(ii + di, jj + dj)
        np.testing.assert_allclose(_haversine_m(lat, lon, la2, lo2) / h, c["dx"] / mf, rtol=2e-5)


def test_namelist_accepts_the_four_projections_and_rejects_others():
    """program_setup.F90:166-191."""
    from mpassit_amd import target_grid as T
    g = T.define_target_grid_params("polar", 101, 91, dx=20000.0, dy=20000.0, ref_lat=80.0, ref_lon=0.0, truelat1=70.0, stand_lon=0.0)
    assert g.proj.code == T.PROJ_PS and g.lat.shape == (90, 100) and g.cosa is None
    g = T.define_target_grid_params("Mercator", 101, 91, dx=20000.0, dy=20000.0, ref_lat=10.0, ref_lon=0.0, truelat1=10.0, stand_lon=0.0)
    assert g.proj.code == T.PROJ_MERC and g.lat_c.shape == (91, 101)
    assert abs(g.lat[44, 49] + g.lat[45, 50] - 20.0) < 0.5           # the known point is the domain centre by default
    with pytest.raises(ValueError):
        T.define_target_grid_params("albers", 101, 91, dx=20000.0, dy=20000.0, ref_lat=10.0, ref_lon=0.0, truelat1=10.0, stand_lon=0.0)
    with pytest.raises(ValueError):
        T.define_target_grid_params("polar", 101, 91, dx=20000.0, dy=20000.0, ref_lat=80.0, ref_lon=0.0, stand_lon=0.0)
