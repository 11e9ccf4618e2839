"""Grid -> Grid destaggering on a GLOBAL lat-lon target (BASELINE config 5: is_regional=.false., wrf_mod_vars).

The reference creates that grid with ESMF_GridCreate1PeriDim(periodicDim=1, poleDim=2, MONOPOLE) (model_grid.F90:685-694):
the CENTER columns wrap and each j end is closed by a pole node whose value is the mean of the neighbouring row.
Parity against the oracle's restatement plus the closed forms a regular lat-lon grid offers: U = mean of the two
zonal neighbours (with wrap, to O(dlon^2)), V between its two meridional neighbours, V at a pole = mean of the row."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-12  # float64 weights, same algorithm on both sides


@pytest.fixture(scope="module")
def global_target():
    from mpassit_amd import target_grid as T
    return T.define_target_grid_params("lat-lon", nx=73, ny=37, stand_lon=0.0, is_regional=False)   # 72 x 36, 5 degrees


@pytest.fixture(scope="module")
def global_grid(gpu_lib, global_target):
    from mpassit_amd import regrid as R
    g = R.Grid.from_target(global_target)
    yield g
    g.destroy()


def _field(t, nlev=3):
    lat, lon = np.deg2rad(t.lat), np.deg2rad(t.lon)
    k = np.arange(1, nlev + 1)[:, None, None]
    return 280.0 + 10.0 * k * np.sin(lat)[None] + 3.0 * np.cos(2 * lon[None] + 0.1 * k) * np.cos(lat)[None]


@pytest.mark.parametrize("stagger", ["EDGE1", "EDGE2"])
def test_weights_match_oracle(oracle, global_target, global_grid, stagger):
    from mpassit_amd import regrid as R
    t = global_target
    st = getattr(R, "STAGGERLOC_" + stagger)
    lon, lat = (t.lon_u, t.lat_u) if stagger == "EDGE1" else (t.lon_v, t.lat_v)
    cen = oracle.lonlat_deg_to_xyz(t.lon, t.lat)
    oi, ow, osrc0, opw = oracle.grid_bilinear_periodic(t.nx, t.ny, 1, cen, st, oracle.lonlat_deg_to_xyz(lon, lat))
    rh = R.regrid_store_grid(global_grid, st)
    gi, gw = rh.weights()
    dst, src0, wp, row_len = rh.pole()
    assert row_len == t.nx and len(dst) == 2 * lon.shape[1]
    cap = opw.reshape(-1) != 0.0
    cap_dst = dst[cap]
    # no unmapped destination on a closed sphere
    assert (gi[:, 0] >= 0).all() and (oi[:, 0] >= 0).all()
    # quads: identical corner ids, weights to rounding; caps: A / B ids + zero-weight fillers on the HIP side
    quad = np.ones(len(gi), bool)
    quad[cap_dst] = False
    assert np.array_equal(gi[quad], oi[quad])
    np.testing.assert_allclose(gw[quad], ow[quad], rtol=0, atol=TOL)
    assert np.array_equal(gi[cap_dst, :2], oi[cap_dst, :2])
    np.testing.assert_allclose(gw[cap_dst, :2], ow[cap_dst, :2], rtol=0, atol=TOL)
    assert (gw[cap_dst, 2:] == 0.0).all() and (oi[cap_dst, 2:] == -1).all()
    np.testing.assert_allclose(wp, opw.reshape(-1), rtol=0, atol=TOL)
    assert np.array_equal(src0[cap], osrc0.reshape(-1)[cap])
    if stagger == "EDGE2":
        assert cap.all() and np.allclose(wp, 1.0, atol=1e-9)      # both V pole rows ARE the pole node
    else:
        assert not cap.any()                                      # U rows coincide with CENTER rows: always in a quad
    rh.release()


def test_destagger_values(oracle, global_target, global_grid):
    from mpassit_amd import regrid as R
    t = global_target
    nlev = 3
    f = _field(t, nlev)
    cen = oracle.lonlat_deg_to_xyz(t.lon, t.lat)
    # U (EDGE1)
    rh = R.regrid_store_grid(global_grid, R.STAGGERLOC_EDGE1)
    u = rh.regrid(f.reshape(-1), nlev=nlev)[0]
    o = oracle.grid_bilinear_periodic(t.nx, t.ny, 1, cen, 1, oracle.lonlat_deg_to_xyz(t.lon_u, t.lat_u))
    np.testing.assert_allclose(u.reshape(nlev, -1), oracle.apply_grid_periodic(t.nx, *o, f, nlev), rtol=TOL, atol=0)
    assert np.array_equal(u[:, :, 0], u[:, :, -1])                 # the duplicate column one period later
    expect = 0.5 * (np.roll(f, 1, axis=2) + f)
    # not exactly 1/2 + 1/2: a latitude circle bulges off the great-circle edge of the CENTER quad, so the row next
    # to it gets a weight of O(dlon^2)
    np.testing.assert_allclose(u[:, :, :-1], expect, rtol=1e-4)
    rh.release()
    # V (EDGE2)
    rh = R.regrid_store_grid(global_grid, R.STAGGERLOC_EDGE2)
    v = rh.regrid(f.reshape(-1), nlev=nlev)[0]
    o = oracle.grid_bilinear_periodic(t.nx, t.ny, 1, cen, 2, oracle.lonlat_deg_to_xyz(t.lon_v, t.lat_v))
    np.testing.assert_allclose(v.reshape(nlev, -1), oracle.apply_grid_periodic(t.nx, *o, f, nlev), rtol=TOL, atol=0)
    np.testing.assert_allclose(v[:, 0], np.broadcast_to(f[:, 0].mean(axis=1)[:, None], v[:, 0].shape), rtol=1e-12)
    np.testing.assert_allclose(v[:, -1], np.broadcast_to(f[:, -1].mean(axis=1)[:, None], v[:, -1].shape), rtol=1e-12)
    # interior V rows: the two meridional neighbours (not exactly 1/2 each: the bilinear map lives on the sphere)
    lo, hi = np.minimum(f[:, :-1], f[:, 1:]), np.maximum(f[:, :-1], f[:, 1:])
    assert (v[:, 1:-1] >= lo - 1e-9).all() and (v[:, 1:-1] <= hi + 1e-9).all()
    # weights sum to one everywhere: a constant field stays constant, poles included
    c = rh.regrid(np.full(t.nx * t.ny, 7.25), nlev=1)[0]
    np.testing.assert_allclose(c, 7.25, rtol=1e-13)
    rh.release()


def test_typed_and_esmf_factor_list(global_target, global_grid):
    """f32 I/O goes through the same pole path; the exported ESMF factor list (pole rows expanded to nx factors of
    w/nx) reproduces the regrid when fed back through mpg_handle_from_weights."""
    import torch

    from mpassit_amd import regrid as R
    t = global_target
    f = _field(t, 2)
    rh = R.regrid_store_grid(global_grid, R.STAGGERLOC_EDGE2)
    ref = rh.regrid(f.reshape(-1), nlev=2)[0]
    src32 = torch.from_numpy(f.astype(np.float32)).cuda()
    out32 = rh.regrid_typed(src32.reshape(-1), nlev=2, out_dtype=torch.float32)
    np.testing.assert_allclose(out32.cpu().numpy().reshape(ref.shape), ref, rtol=3e-7)
    row, col, S = rh.to_esmf_weights()
    per_row = np.bincount(row - 1, minlength=rh.n_dst)[: t.nx]                  # south pole row: nx factors for the pole
    assert ((per_row >= t.nx) & (per_row <= t.nx + 2)).all()                    # node (+ rounding-level A / B entries)
    byo = R.RouteHandle.from_weights(t.nx * t.ny, t.nx, t.ny + 1, row, col, S)
    np.testing.assert_allclose(byo.regrid(f.reshape(-1), nlev=2)[0].reshape(ref.shape), ref, rtol=1e-13)
    byo.release()
    rh.release()


def test_row_blocks_reproduce_the_global_result(global_target, global_grid):
    """Row sharding (dist.ShardedInterp): each block extended by one halo row, caps only where the block touches a pole."""
    from mpassit_amd import regrid as R
    t = global_target
    nlev = 2
    f = _field(t, nlev)
    for st in (R.STAGGERLOC_EDGE1, R.STAGGERLOC_EDGE2):
        rh = R.regrid_store_grid(global_grid, st)
        full = rh.regrid(f.reshape(-1), nlev=nlev)[0]
        rh.release()
        pieces = []
        for j0, j1 in ((0, 13), (13, 25), (25, t.ny)):
            e0, e1 = max(j0 - 1, 0), min(j1 + 1, t.ny)
            g = R.Grid.from_target(t, rows=(e0, e1))
            rb = R.regrid_store_grid(g, st)
            blk = rb.regrid(np.ascontiguousarray(f[:, e0:e1]).reshape(-1), nlev=nlev)[0]
            n = j1 - j0 + (1 if (st == R.STAGGERLOC_EDGE2 and j1 == t.ny) else 0)
            pieces.append(blk[:, j0 - e0:j0 - e0 + n])
            rb.release()
            g.destroy()
        assert np.array_equal(np.concatenate(pieces, axis=1), full)


def test_periodic_flag_validation(gpu_lib):
    from mpassit_amd import regrid as R
    lon, lat = np.meshgrid(np.arange(4.0), np.arange(3.0))
    with pytest.raises(gpu_lib.MpgError):
        R.Grid(lon, lat, periodic=2)        # a cap switch without MPG_GRID_PERIODIC_I


def test_conservative_with_corner_latitudes_rounded_past_the_pole(oracle, gpu_lib, global_target):
    """-89.95 + 1800 * 0.1 is not exactly 90 in floating point: when a CORNER latitude lands an ulp beyond the pole,
    cos(lat) changes sign and the collapsed top side of the polar cells points the other way.  It must be ignored
    (seen at the full config-5 size, where the whole northern row came out uncovered)."""
    import copy

    from mpassit_amd import regrid as R, synth
    t = copy.copy(global_target)
    t.lat_c = t.lat_c.copy()
    t.lat_c[-1] = np.nextafter(90.0, 91.0)
    t.lat_c[0] = -np.nextafter(90.0, 91.0)
    assert np.cos(np.deg2rad(t.lat_c[-1, 0])) < 0.0
    m = synth.icosahedral_mesh(4)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(t)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_CONSERVE)
    rp, col, val = rh.csr()
    P = t.nx * t.ny
    rowsum = np.bincount(np.repeat(np.arange(P), np.diff(rp)), weights=val, minlength=P)
    assert np.abs(rowsum - 1).max() < 1e-9
    vlon, vlat = oracle.mesh_coords_deg(m.lonVertex, m.latVertex)
    rp_o, col_o, val_o = oracle.conserve(m.verticesOnCell, oracle.lonlat_deg_to_xyz(vlon, vlat), t.nx, t.ny,
                                         oracle.lonlat_deg_to_xyz(t.lon_c, t.lat_c))
    snow = synth.snow_field(m.latCell, m.lonCell) + 1.0
    want = oracle.apply_csr(rp_o, col_o, val_o, snow, 1)[0]
    np.testing.assert_allclose(rh.regrid(snow, nlev=1).reshape(-1), want, rtol=1e-10)
    rh.release()
    mesh.destroy()
    grid.destroy()
