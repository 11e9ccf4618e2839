"""BASELINE config 5 at full size: 3 003 042-cell global geodesic mesh ("3 M-cell", round 6) -> 3600 x 1800 global lat-lon grid
(is_regional=.false.), wrf_mod_vars staggered winds.  Properties only a GLOBAL remap offers -- every target point is
mapped (poles and date line included), the conservative remap preserves the global integral, the pole rows of V are
zonal means, U wraps around the seam -- and the oracle itself at full size (test_c5_full_size_oracle_parity: the three
weight sets entry by entry over all 6.48 M points)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c5(gpu_lib):
    import torch
    from mpassit_amd import regrid as R, workloads
    m, g, nlev, _ = workloads.workload("c5_global_latlon")
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    yield dict(m=m, g=g, mesh=mesh, grid=grid, nlev=nlev, torch=torch)
    mesh.destroy()
    grid.destroy()


def _xyz(lat, lon):
    cl = np.cos(lat)
    return np.stack([cl * np.cos(lon), cl * np.sin(lon), np.sin(lat)], axis=-1)


def _tri_area(a, b, c):
    """Spherical triangle area (Van Oosterom-Strackee), difference form for small triangles."""
    num = np.einsum("ij,ij->i", a, np.cross(b - a, c - a))
    den = 1.0 + np.einsum("ij,ij->i", a, b) + np.einsum("ij,ij->i", b, c) + np.einsum("ij,ij->i", c, a)
    return 2.0 * np.arctan2(num, den)


def test_c5_bilinear_and_nearest(c5):
    from mpassit_amd import regrid as R
    torch, m, g = c5["torch"], c5["m"], c5["g"]
    assert (g.nx, g.ny) == (3600, 1800) and m.nCells == 3_003_042 >= 3_000_000      # "3 M-cell mesh" (BASELINE configs[4])
    rh = R.regrid_store(c5["mesh"], c5["grid"], R.REGRIDMETHOD_BILINEAR)
    idx, w = rh.weights()
    assert (idx >= 0).all()                                # a closed sphere has no unmapped point
    assert np.abs(w.sum(1) - 1).max() < 1e-13 and w.min() > -1e-10
    cx = _xyz(m.latCell, np.where(m.lonCell > np.pi, m.lonCell - 2 * np.pi, m.lonCell))
    px = _xyz(np.deg2rad(g.lat).ravel(), np.deg2rad(g.lon).ravel())
    a = np.array([0.3, -0.7, 0.5])
    src = torch.as_tensor(np.stack([cx @ a, np.full(m.nCells, 7.25)]), device="cuda")
    out = rh.regrid(src.view(-1), nlev=2).cpu().numpy().reshape(2, -1)
    A, B, C = cx[idx[:, 0]], cx[idx[:, 1]], cx[idx[:, 2]]
    n = np.cross(B - A, C - A)
    t = np.einsum("ij,ij->i", A, n) / np.einsum("ij,ij->i", n, px)       # closed form, SURVEY App. E
    assert np.abs(out[0] - t * (px @ a)).max() < 1e-12
    assert np.abs(out[1] - 7.25).max() < 1e-13
    print("store_ms bilinear C5:", rh.store_ms)
    rh.release()
    rn = R.regrid_store(c5["mesh"], c5["grid"], R.REGRIDMETHOD_NEAREST_STOD)
    near = rn.weights()[0][:, 0]
    assert near.min() >= 0
    # the nearest cell is a corner of the containing Delaunay triangle or beats all three of them
    d_near = ((cx[near] - px) ** 2).sum(1)
    d_tri = np.minimum(np.minimum(((A - px) ** 2).sum(1), ((B - px) ** 2).sum(1)), ((C - px) ** 2).sum(1))
    assert (d_near <= d_tri * (1 + 1e-12)).all()
    for p in np.random.default_rng(5).choice(px.shape[0], 25, replace=False):   # brute force over all cells
        d = ((cx - px[p]) ** 2).sum(1)
        assert d[near[p]] <= d.min() * (1 + 1e-12)
    print("store_ms nearest C5:", rn.store_ms)
    rn.release()


def test_c5_conservative_global_integral(c5):
    from mpassit_amd import regrid as R, synth
    torch, m, g = c5["torch"], c5["m"], c5["g"]
    rh = R.regrid_store(c5["mesh"], c5["grid"], R.REGRIDMETHOD_CONSERVE)
    rp, col, val = rh.csr()
    P = g.nx * g.ny
    rows = np.repeat(np.arange(P), np.diff(rp))
    rowsum = np.bincount(rows, weights=val, minlength=P)
    assert val.min() > 0 and np.abs(rowsum - 1).max() < 1e-9            # every target cell fully covered
    # areas of the target cells (great-circle quads between the CORNER points, as the clipper sees them)
    cor = _xyz(np.deg2rad(g.lat_c), np.deg2rad(g.lon_c))
    c00, c10, c11, c01 = (cor[:-1, :-1].reshape(-1, 3), cor[:-1, 1:].reshape(-1, 3), cor[1:, 1:].reshape(-1, 3),
                          cor[1:, :-1].reshape(-1, 3))
    a_dst = _tri_area(c00, c10, c11) + _tri_area(c00, c11, c01)
    assert abs(a_dst.sum() - 4 * np.pi) < 1e-9
    # column sums: sum_d A_d w_ds = A_s, the area of source cell s (every source cell lies inside the global grid)
    a_src = np.bincount(col, weights=val * a_dst[rows], minlength=m.nCells)
    assert abs(a_src.sum() - 4 * np.pi) < 1e-8
    vx = _xyz(m.latVertex, m.lonVertex)
    voc = m.verticesOnCell
    v0 = vx[voc[:, 0] - 1]
    poly = np.zeros(m.nCells)
    for k in range(1, voc.shape[1] - 1):
        ok = voc[:, k + 1] > 0
        poly[ok] += _tri_area(v0[ok], vx[voc[ok, k] - 1], vx[voc[ok, k + 1] - 1])
    assert np.abs(a_src / poly - 1).max() < 1e-7
    # hence the global integral of a remapped field is preserved
    f = synth.snow_field(m.latCell, m.lonCell) + 1.0
    out = rh.regrid(torch.as_tensor(f, device="cuda"), nlev=1).cpu().numpy().ravel()
    assert abs((out * a_dst).sum() / (f * poly).sum() - 1) < 1e-9
    print("store_ms conserve C5:", rh.store_ms, "nnz/row", len(col) / P)
    rh.release()


def test_c5_staggered_winds(c5):
    """wrf_mod_vars: UMASS/VMASS on CENTER -> U on EDGE1, V on EDGE2 of the periodic grid (interp.F90:295-328)."""
    from mpassit_amd import regrid as R
    torch, g = c5["torch"], c5["g"]
    nlev = 4
    lat, lon = np.deg2rad(g.lat), np.deg2rad(g.lon)
    mass = np.stack([10.0 * (k + 1) * np.cos(lat) * np.sin(lon + 0.3 * k) + 2.0 * np.sin(lat) for k in range(nlev)])
    src = torch.as_tensor(mass, device="cuda")
    ru = R.regrid_store_grid(c5["grid"], R.STAGGERLOC_EDGE1)
    u = ru.regrid(src.view(-1), nlev=nlev)[0].cpu().numpy()
    assert u.shape == (nlev, g.ny, g.nx + 1)
    assert np.array_equal(u[:, :, 0], u[:, :, -1])
    np.testing.assert_allclose(u[:, :, :-1], 0.5 * (np.roll(mass, 1, axis=2) + mass), atol=1e-4)
    ru.release()
    rv = R.regrid_store_grid(c5["grid"], R.STAGGERLOC_EDGE2)
    v = rv.regrid(src.view(-1), nlev=nlev)[0].cpu().numpy()
    assert v.shape == (nlev, g.ny + 1, g.nx)
    np.testing.assert_allclose(v[:, 0], np.broadcast_to(mass[:, 0].mean(axis=1)[:, None], v[:, 0].shape), atol=1e-12)
    np.testing.assert_allclose(v[:, -1], np.broadcast_to(mass[:, -1].mean(axis=1)[:, None], v[:, -1].shape), atol=1e-12)
    np.testing.assert_allclose(v[:, 1:-1], 0.5 * (mass[:, :-1] + mass[:, 1:]), atol=1e-5)
    one = rv.regrid(torch.full((g.nx * g.ny,), 3.5, dtype=torch.float64, device="cuda"), nlev=1)
    assert float((one - 3.5).abs().max()) < 1e-13
    rv.release()


def test_c5_full_size_oracle_parity(c5, oracle):
    """configs[4] against the ORACLE at full size (2.6 M cells -> 6.48 M points, poles and date line included): the three
    weight sets entry by entry, every differing entry examined, and a 5-level field through the default Regrid."""
    from _parity_helpers import assert_csr_equal, assert_fixed_weights_equal, assert_nearest_equal, conserve_tol, rel_err
    from conftest import mesh_xyz
    from mpassit_amd import regrid as R, synth
    torch, m, g = c5["torch"], c5["m"], c5["g"]
    o = oracle
    cxyz, vxyz = mesh_xyz(o, m)
    tri, _ = o.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    assert np.array_equal(c5["mesh"].triangles(), tri)
    pxyz = o.lonlat_deg_to_xyz(g.lon, g.lat)
    idx_o, w_o = o.bilinear_weights(cxyz, tri, pxyz)
    rh = R.regrid_store(c5["mesh"], c5["grid"], R.REGRIDMETHOD_BILINEAR)
    idx_g, w_g = rh.weights()
    n_ties = assert_fixed_weights_equal(idx_o, w_o, idx_g, w_g, tol=1e-10)
    assert n_ties <= 1e-4 * idx_o.shape[0]
    L = 5
    src = synth.analytic_field(m.latCell, m.lonCell, L)
    got = rh.regrid(torch.as_tensor(src, device="cuda").view(-1), nlev=L).cpu().numpy().reshape(L, -1)
    assert rel_err(got, o.apply_fixed(idx_o, w_o, src, L)) < 1e-11 < 1e-6
    rh.release()
    del idx_o, w_o, idx_g, w_g, got
    rh = R.regrid_store(c5["mesh"], c5["grid"], R.REGRIDMETHOD_NEAREST_STOD)
    # rows at +-89.95 degrees: 3600 points within 5 km of the pole see several cells at nearly equal distance
    assert_nearest_equal(o.nearest(cxyz, pxyz), rh.weights()[0][:, 0], pxyz, cxyz, max_ties=16)
    rh.release()
    rh = R.regrid_store(c5["mesh"], c5["grid"], R.REGRIDMETHOD_CONSERVE)
    rp_o, col_o, val_o = o.conserve(m.verticesOnCell, vxyz, g.nx, g.ny, o.lonlat_deg_to_xyz(g.lon_c, g.lat_c))
    rp_g, col_g, val_g = rh.csr()
    # polar slivers: a 0.1-degree cell at 89.95 degrees is 11 km x 10 m -> area 4e-12 of the unit sphere; weights there are
    # ratios of areas that carry ~1e-16 absolute rounding, so the bound of conserve_tol reaches 1e-3 for those rows and
    # is no test.  Compare WEIGHTED by destination area instead (what conservation and the regridded values see): the
    # absolute area of every overlap A_j w_ij agrees to 1e-13 of the unit sphere, all latitudes alike.
    lat_e = np.deg2rad(g.lat_c[:, 0])
    A_dst = np.repeat(np.deg2rad(360.0 / g.nx) * (np.sin(lat_e[1:]) - np.sin(lat_e[:-1])), g.nx)
    rows_o = np.repeat(np.arange(A_dst.size), np.diff(rp_o))
    rows_g = np.repeat(np.arange(A_dst.size), np.diff(rp_g))
    n_common, only_o, only_g = assert_csr_equal(rp_o, col_o, val_o * A_dst[rows_o], rp_g, col_g, val_g * A_dst[rows_g], m.nCells,
                                                tol=1e-13, sliver=1e-13)
    assert n_common > 0.99 * max(col_o.size, col_g.size)
    # away from the polar caps (|lat| < 85) the plain weights obey the area-scaled bound as everywhere else
    mid_rows = np.abs(np.repeat(g.lat[:, 0], g.nx)) < 85.0
    keep_o, keep_g = mid_rows[rows_o], mid_rows[rows_g]
    tol = max(1e-11, 16 * np.finfo(np.float64).eps / A_dst[mid_rows].min())

    def sub(rp, col, val, keep, rows):
        cnt = np.bincount(rows[keep], minlength=A_dst.size)
        return np.concatenate([[0], np.cumsum(cnt)]), col[keep], val[keep]
    assert_csr_equal(*sub(rp_o, col_o, val_o, keep_o, rows_o), *sub(rp_g, col_g, val_g, keep_g, rows_g), m.nCells, tol=tol)
    snow = synth.snow_field(m.latCell, m.lonCell, lat0=np.deg2rad(80.0))
    got = rh.regrid(snow, nlev=1).reshape(-1)
    want = o.apply_csr(rp_o, col_o, val_o, snow, 1)[0]
    assert np.abs(got - want).max() < 1e-9 * np.abs(want).max() < 1e-6
    rh.release()


def test_c5_hundred_field_bundle_as_baseline_states_it(c5, oracle):
    """BASELINE configs[4] as written: "3 M-cell mesh -> 3600 x 1800 global lat-lon, 100+ 3-D fields" -- ONE bundle Regrid of
    100 float32 file-order fields x 55 levels (interp.F90:240-254 regrids every listed nz field in one FieldBundleRegrid):
    66.1 GB of sources, 142.6 GB of results, 1.3 M workgroups.  (The mesh: the class-I geodesic grid of frequency 548, 3 003 042 cells;
    until round 5 the 2 621 442-cell bisection mesh stood in.)
    What is checked: the first and the LAST field of the bundle -- where a 32-bit overflow in field * level * point offsets
    would land -- against the oracle's widen -> apply chain narrowed to float32 (equal except where the two float64 values,
    <= 1e-12 apart, straddle a float32 rounding boundary: one float32 ulp on a vanishing fraction of the points, the bar of
    tests/test_typed_oracle_gpu.py), a constant field in the middle, and that no field is left unwritten."""
    from mpassit_amd import regrid as R
    torch, m, g, nlev = c5["torch"], c5["m"], c5["g"], c5["nlev"]
    F = 100
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    need = F * nlev * 4 * (m.nCells + g.nx * g.ny) + 8e9
    if free < need:
        pytest.skip("needs %.0f GB of free HBM, %.0f free" % (need / 1e9, free / 1e9))
    rh = R.regrid_store(c5["mesh"], c5["grid"], R.REGRIDMETHOD_BILINEAR)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    src = torch.empty((F, m.nCells, nlev), dtype=torch.float32, device="cuda")
    for f in range(F):
        src[f].uniform_(-1.0, 1.0, generator=gen)
    src[F // 2].fill_(3.25)
    out = torch.full((F, nlev, g.ny, g.nx), float("nan"), dtype=torch.float32, device="cuda")
    rh.regrid_typed(src.view(-1), nlev=nlev, nfields=F, layout=R.LAYOUT_LEV_FAST, out=out)
    torch.cuda.synchronize()
    assert rh.kernel_choice()[1] > 0                               # the staged file-order kernel (k_apply3_lfu) serves configuration 5
    for f in range(F):
        assert not torch.isnan(out[f, ::18, ::60, ::60]).any(), f   # a sample of every field: all written
    assert not torch.isnan(out[F - 1]).any() and not torch.isnan(out[0]).any()
    assert bool((out[F // 2] == 3.25).all())                       # sum of the weights is 1 on a closed sphere; float32(3.25) exact
    idx, w = rh.weights()
    for f in (0, F - 1):
        want = oracle.apply_fixed(idx, w, src[f].cpu().numpy().astype(np.float64), nlev, lev_fast=True).astype(np.float32)
        got = out[f].cpu().numpy().reshape(nlev, -1)
        ne = got != want
        assert ne.mean() < 1e-5, (f, ne.mean())
        if ne.any():
            assert (np.abs(got[ne].astype(np.float64) - want[ne]) <= np.spacing(np.abs(want[ne])).astype(np.float64)).all(), f
    rh.release()
    del src, out
    torch.cuda.empty_cache()
