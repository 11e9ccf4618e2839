"""CPU tests of the oracle: pinned against the compiled-reference projection goldens and the
analytic known-answer tests that stand in for the (non-existent) reference test-suite."""
import json
import os

import numpy as np
import pytest

from conftest import mesh_xyz

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "projection_lc.json")))


def test_projection_goldens_oracle(oracle):
    nl = GOLD["namelist"]
    p = oracle.Proj.lambert(nl["truelat1"], nl["truelat2"], nl["stand_lon"], nl["ref_lat"], nl["ref_lon"], GOLD["knowni"], GOLD["knownj"], nl["dx"])
    f = p.fields()
    for k in ("cone", "rsw", "polei", "polej"):
        assert f[k] == pytest.approx(GOLD[k], rel=1e-15, abs=0)
    assert p.xytoll(1, 1, oracle.M) == pytest.approx(GOLD["mass_1_1"], rel=1e-14)
    assert p.xytoll(1800, 1060, oracle.M) == pytest.approx(GOLD["mass_1800_1060"], rel=1e-14)
    assert p.xytoll(1, 1, oracle.CORNER) == pytest.approx(GOLD["corner_1_1"], rel=1e-14)
    assert p.xytoll(900.5, 530.5, oracle.M) == pytest.approx(GOLD["xy_900p5_530p5"], rel=1e-14)
    assert p.latlon_to_ij(38.5, -97.5) == pytest.approx(GOLD["latlon_to_ij_38p5_m97p5"], rel=1e-14)


def test_projection_goldens_host_mirror():
    """mpassit_amd.target_grid (product host code) reproduces the same compiled-reference values."""
    from mpassit_amd import target_grid as tg
    nl = GOLD["namelist"]
    g = tg.define_target_grid_params("lambert", nl["nx"], nl["ny"], dx=nl["dx"], dy=nl["dy"], ref_lat=nl["ref_lat"], ref_lon=nl["ref_lon"],
                                     truelat1=nl["truelat1"], truelat2=nl["truelat2"], stand_lon=nl["stand_lon"])
    assert (g.nx, g.ny) == (GOLD["i_target"], GOLD["j_target"])           # program_setup.F90:163-164
    assert (g.proj.knowni, g.proj.knownj) == (GOLD["knowni"], GOLD["knownj"])  # default known point = centre (:238-240)
    for k in ("cone", "rsw", "polei", "polej"):
        assert getattr(g.proj, k) == pytest.approx(GOLD[k], rel=1e-15)
    assert (g.lat[0, 0], g.lon[0, 0]) == pytest.approx(GOLD["mass_1_1"], rel=1e-14)
    assert (g.lat[-1, -1], g.lon[-1, -1]) == pytest.approx(GOLD["mass_1800_1060"], rel=1e-14)
    assert (g.lat_c[0, 0], g.lon_c[0, 0]) == pytest.approx(GOLD["corner_1_1"], rel=1e-14)
    assert g.lat_u.shape == (1060, 1801) and g.lat_v.shape == (1061, 1800) and g.lat_c.shape == (1061, 1801)


def test_target_grid_host_mirror_equals_oracle(oracle, conus_grid_30km):
    from mpassit_amd import target_grid as tg
    g = conus_grid_30km
    p = g.proj
    po = oracle.Proj.lambert(p.truelat1, p.truelat2, p.stdlon, p.lat1, p.lon1, p.knowni, p.knownj, p.dx)
    for st_o, lat, lon in ((oracle.M, g.lat, g.lon), (oracle.U, g.lat_u, g.lon_u), (oracle.V, g.lat_v, g.lon_v), (oracle.CORNER, g.lat_c, g.lon_c)):
        la, lo = po.lat_lon_fields(lat.shape[1], lat.shape[0], st_o)
        assert np.abs(la - lat).max() < 1e-12 and np.abs(lo - lon).max() < 1e-12
    ca, sa = oracle.get_rotang(g.lat, g.lon)
    assert np.abs(ca - g.cosa).max() < 1e-13 and np.abs(sa - g.sina).max() < 1e-13
    # lat-lon (global) branch of read_setup_namelist: program_setup.F90:197-210
    gl = tg.define_target_grid_params("lat-lon", 361, 181, stand_lon=0.0, is_regional=False)
    assert (gl.nx, gl.ny) == (360, 180)
    assert gl.lat[0, 0] == pytest.approx(-89.5) and gl.lon[0, 0] == pytest.approx(0.5) and gl.lon[0, -1] == pytest.approx(359.5)
    pl = oracle.Proj.latlon(gl.proj.lat1, gl.proj.lon1, 1.0, 1.0, gl.proj.latinc, gl.proj.loninc)
    la, lo = pl.lat_lon_fields(360, 180, oracle.M)
    assert np.array_equal(la, gl.lat) and np.array_equal(lo, gl.lon)


def test_para_range(oracle):
    from mpassit_amd import dist
    for n, p in ((10, 3), (655362, 120), (7, 8), (100, 1)):
        got = [oracle.para_range(1, n, p, r) for r in range(p)]
        assert got == [dist.para_range(1, n, p, r) for r in range(p)]
        sizes = [b - a + 1 for a, b in got]
        assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
        assert sizes[: n % p] == [n // p + 1] * (n % p)  # first mod(N,p) ranks get the extra (App. E)
        assert got[0][0] == 1 and got[-1][1] == n and all(got[i][1] + 1 == got[i + 1][0] for i in range(p - 1))


def test_mesh_coords_wrap(oracle):
    lon = np.array([0.0, np.pi, np.pi + 1e-9, 2 * np.pi - 1e-3, 1.0])
    lo, la = oracle.mesh_coords_deg(lon, np.zeros(5))
    assert lo[0] == 0 and lo[1] == pytest.approx(180.0) and lo[2] < -179.9 and lo[3] < 0 and lo[4] == pytest.approx(np.degrees(1.0))


def test_dual_triangles(oracle, global_mesh, regional_case):
    m = global_mesh
    cxyz, vxyz = mesh_xyz(oracle, m)
    tri, nvalid = oracle.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    assert nvalid == m.nVertices == 2 * m.nCells - 4      # closed sphere: every vertex has 3 cells (Euler)
    A, B, C = cxyz[tri[:, 0]], cxyz[tri[:, 1]], cxyz[tri[:, 2]]
    assert (np.einsum("ij,ij->i", A, np.cross(B, C)) > 0).all()   # CCW seen from outside
    # the vertex is the circumcentre of its triangle (Voronoi/Delaunay duality)
    assert max(np.abs(np.linalg.norm(vxyz - A, axis=1) - np.linalg.norm(vxyz - B, axis=1)).max(),
               np.abs(np.linalg.norm(vxyz - A, axis=1) - np.linalg.norm(vxyz - C, axis=1)).max()) < 1e-9
    mr, _ = regional_case
    cr, _ = mesh_xyz(oracle, mr)
    tri_r, nv_r = oracle.dual_triangles(mr.verticesOnCell, mr.nVertices, cr)
    assert 0 < nv_r < mr.nVertices        # rim vertices touch < 3 kept cells -> no dual element (App. A2)
    assert ((tri_r >= 0).all(axis=1) | (tri_r < 0).all(axis=1)).all()


def test_bilinear_known_answers(oracle, global_mesh, conus_grid_30km):
    m, g = global_mesh, conus_grid_30km
    cxyz, _ = mesh_xyz(oracle, m)
    tri, _ = oracle.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    pxyz = oracle.lonlat_deg_to_xyz(g.lon, g.lat)
    idx, w = oracle.bilinear_weights(cxyz, tri, pxyz)
    assert (idx >= 0).all()
    assert np.abs(w.sum(1) - 1).max() < 1e-14 and w.min() >= -1e-10
    # constant field -> exactly the constant (sum w = 1)
    c = oracle.apply_fixed(idx, w, np.full((2, m.nCells), 7.5), 2)
    assert np.abs(c - 7.5).max() < 1e-13
    # field linear in Cartesian X: result = t * (a.P), t = det[A,B,C]/S (App. A2 / App. E)
    a = np.array([0.3, -0.7, 0.5])
    got = oracle.apply_fixed(idx, w, (cxyz @ a)[None, :], 1)[0]
    A, B, C = cxyz[idx[:, 0]], cxyz[idx[:, 1]], cxyz[idx[:, 2]]
    t = np.einsum("ij,ij->i", A, np.cross(B, C)) / np.einsum("ij,ij->i", np.cross(B - A, C - A), pxyz)
    assert np.abs(got - t * (pxyz @ a)).max() < 1e-13
    # brute-force containment check on a sample: the chosen triangle is the lowest-id containing one
    rng = np.random.default_rng(0)
    ok = tri[:, 0] >= 0
    for p in rng.choice(len(pxyz), 25, replace=False):
        P = pxyz[p]
        TA, TB, TC = cxyz[tri[ok, 0]] - P, cxyz[tri[ok, 1]] - P, cxyz[tri[ok, 2]] - P
        d = np.stack([np.cross(TB, TC) @ P, np.cross(TC, TA) @ P, np.cross(TA, TB) @ P], axis=1)
        S = d.sum(1)
        inside = (S > 0) & ((d / np.where(S > 0, S, 1)[:, None]) >= -1e-10).all(1)
        first = np.nonzero(ok)[0][np.argmax(inside)]
        assert sorted(tri[first]) == sorted(idx[p])
    # level-fastest source order gives the same answer
    src = np.random.default_rng(1).standard_normal((4, m.nCells))
    assert np.array_equal(oracle.apply_fixed(idx, w, src, 4), oracle.apply_fixed(idx, w, np.ascontiguousarray(src.T), 4, lev_fast=True))


def test_bilinear_unmapped_outside_hull(oracle, regional_case):
    m, g = regional_case
    cxyz, _ = mesh_xyz(oracle, m)
    tri, _ = oracle.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    idx, w = oracle.bilinear_weights(cxyz, tri, oracle.lonlat_deg_to_xyz(g.lon, g.lat))
    un = (idx[:, 0] < 0).reshape(g.ny, g.nx)
    # the mesh covers index range [1,141]x[1,81] of the 150x90 target: the north and east strips lie outside the hull
    assert un[-1, :].all() and un[:, -1].all() and un[-5:, :].all() and un[:, -5:].all()
    assert not un[g.ny // 2, g.nx // 2]
    out = oracle.apply_fixed(idx, w, np.ones((1, m.nCells)), 1).reshape(g.ny, g.nx)
    assert (out[un] == 0.0).all() and np.abs(out[~un] - 1).max() < 1e-14    # zero-filled outside (slide 16)


def test_nearest_matches_brute_force_and_ties(oracle, global_mesh, conus_grid_30km, regional_case):
    m, g = global_mesh, conus_grid_30km
    cxyz, _ = mesh_xyz(oracle, m)
    pxyz = oracle.lonlat_deg_to_xyz(g.lon, g.lat)
    assert np.array_equal(oracle.nearest(cxyz, pxyz[::7]), oracle.nearest(cxyz, pxyz[::7], brute=True))
    mr, gr = regional_case            # includes target points far outside the mesh footprint: all mapped (App. A6)
    cr, _ = mesh_xyz(oracle, mr)
    pr = oracle.lonlat_deg_to_xyz(gr.lon, gr.lat)
    ir = oracle.nearest(cr, pr)
    assert (ir >= 0).all() and np.array_equal(ir[::11], oracle.nearest(cr, pr[::11], brute=True))
    # exact tie -> lowest cell id: sites mirrored about the plane y = 0, query on the plane
    sites = np.array([[0.6, 0.3, np.sqrt(1 - 0.45)], [0.6, -0.3, np.sqrt(1 - 0.45)], [-1.0, 0.0, 0.0]])
    q = np.array([[np.sqrt(0.5), 0.0, np.sqrt(0.5)]])
    assert oracle.nearest(sites, q)[0] == 0 and oracle.nearest(sites[[1, 0, 2]], q)[0] == 0
    # nearest-neighbour regrid is a pure copy
    from mpassit_amd import synth
    cat = synth.category_field(mr.nCells, nlev=3)
    out = oracle.apply_nearest(ir, cat, 3)
    assert np.array_equal(out, cat[:, ir])


def test_conservative_known_answers(oracle, global_mesh, conus_grid_30km, regional_case):
    m, g = global_mesh, conus_grid_30km
    _, vxyz = mesh_xyz(oracle, m)
    pc = oracle.lonlat_deg_to_xyz(g.lon_c, g.lat_c)
    rp, col, val = oracle.conserve(m.verticesOnCell, vxyz, g.nx, g.ny, pc)
    P = g.nx * g.ny
    rows = np.repeat(np.arange(P), np.diff(rp))
    assert val.min() > 0
    rowsum = np.bincount(rows, weights=val, minlength=P)
    assert np.abs(rowsum - 1).max() < 1e-10                        # fully covered cells: weights sum to 1
    assert np.abs(oracle.apply_csr(rp, col, val, np.full((1, m.nCells), 4.0), 1) - 4.0).max() < 1e-9
    # conservation: sum_j dst_j A_j = sum_i src_i A_i^overlap  (App. E), with A from the same weights
    cq = pc.reshape(g.ny + 1, g.nx + 1, 3)

    def tri_area(a, b, c):
        num = np.einsum("...i,...i->...", a, np.cross(b - a, c - a))
        den = 1 + np.einsum("...i,...i->...", a, b) + np.einsum("...i,...i->...", b, c) + np.einsum("...i,...i->...", c, a)
        return 2 * np.arctan2(num, den)
    Aq = (tri_area(cq[:-1, :-1], cq[:-1, 1:], cq[1:, 1:]) + tri_area(cq[:-1, :-1], cq[1:, 1:], cq[1:, :-1])).ravel()
    src = np.random.default_rng(2).uniform(0, 3, m.nCells)
    dst = oracle.apply_csr(rp, col, val, src[None, :], 1)[0]
    overlap = np.bincount(col, weights=val * Aq[rows], minlength=m.nCells)     # area of src_i inside the target domain
    assert abs((dst * Aq).sum() - (src * overlap).sum()) < 1e-12 * (dst * Aq).sum() * 10
    # regional mesh smaller than the target: partially covered / uncovered destination cells
    mr, gr = regional_case
    _, vr = mesh_xyz(oracle, mr)
    rp2, col2, val2 = oracle.conserve(mr.verticesOnCell, vr, gr.nx, gr.ny, oracle.lonlat_deg_to_xyz(gr.lon_c, gr.lat_c))
    rs2 = np.bincount(np.repeat(np.arange(gr.nx * gr.ny), np.diff(rp2)), weights=val2, minlength=gr.nx * gr.ny)
    assert rs2.max() < 1 + 1e-10 and rs2.min() == 0.0 and ((rs2 > 1e-3) & (rs2 < 0.999)).any()


def test_grid_to_grid_bilinear(oracle, conus_grid_30km):
    g = conus_grid_30km
    cxyz = oracle.lonlat_deg_to_xyz(g.lon, g.lat)
    idx, w = oracle.grid_bilinear(g.nx, g.ny, cxyz, 1, oracle.lonlat_deg_to_xyz(g.lon_u, g.lat_u))
    idx = idx.reshape(g.ny, g.nx + 1, 4)
    w = w.reshape(g.ny, g.nx + 1, 4)
    assert (idx[:, 0, 0] < 0).all() and (idx[:, -1, 0] < 0).all()      # outer half-cell ring unmapped (App. A4)
    inner = idx[1:-1, 1:-1]
    assert (inner[..., 0] >= 0).all()
    ws = w[1:-1, 1:-1]
    assert np.abs(ws.sum(-1) - 1).max() < 1e-12
    # a U point sits (to sphere-vs-plane accuracy) midway between two mass points: two weights ~ 1/2
    top2 = np.sort(ws, axis=-1)[..., 2:]
    assert np.abs(top2 - 0.5).max() < 1e-3


def test_rotation_identity(oracle, conus_grid_30km):
    g = conus_grid_30km
    rng = np.random.default_rng(4)
    u, v = rng.standard_normal((3, g.ny, g.nx)), rng.standard_normal((3, g.ny, g.nx))
    un, vn = oracle.rotate_winds(g.cosa, g.sina, u, v)
    assert np.abs(un - (u * g.cosa + v * g.sina)).max() < 1e-12      # interp.F90:741-742 == rotation by alpha
    assert np.abs(vn - (v * g.cosa - u * g.sina)).max() < 1e-12
    assert np.abs(g.cosa ** 2 + g.sina ** 2 - 1).max() < 1e-14


def test_node_located_bilinear_fan(oracle, global_mesh, conus_grid_30km):
    """App. A3 (vorticity): values on Voronoi vertices, fan triangulation from the first listed vertex."""
    m, g = global_mesh, conus_grid_30km
    _, vxyz = mesh_xyz(oracle, m)
    ftri, nvalid = oracle.fan_triangles(m.verticesOnCell, vxyz)
    ne = (m.verticesOnCell > 0).sum(1)
    assert nvalid == (ne - 2).sum()                       # an n-gon gives n-2 fan triangles
    pxyz = oracle.lonlat_deg_to_xyz(g.lon, g.lat)
    idx, w = oracle.bilinear_weights(vxyz, ftri, pxyz)
    assert (idx >= 0).all() and np.abs(w.sum(1) - 1).max() < 1e-13
    # the three vertices belong to the Voronoi cell of the nearest cell centre (polygon containment)
    cxyz, _ = mesh_xyz(oracle, m)
    near = oracle.nearest(cxyz, pxyz)
    cell_verts = m.verticesOnCell[near] - 1
    inside = np.array([(np.isin(idx[p], cell_verts[p])).all() for p in range(len(idx))])
    assert inside.mean() > 0.999                          # shared-edge ties may pick the neighbouring polygon
    a = np.array([0.2, 0.5, -0.4])
    got = oracle.apply_fixed(idx, w, (vxyz @ a)[None, :], 1)[0]
    A, B, C = vxyz[idx[:, 0]], vxyz[idx[:, 1]], vxyz[idx[:, 2]]
    n = np.cross(B - A, C - A)
    t = np.einsum("ij,ij->i", A, n) / np.einsum("ij,ij->i", n, pxyz)
    assert np.abs(got - t * (pxyz @ a)).max() < 1e-13


def test_icosahedral_mesh_is_a_valid_mpas_mesh(oracle):
    """synth.icosahedral_mesh: 10*4^L+2 cells, 12 pentagons, every Voronoi vertex has 3 cells, polygons CCW and tiling."""
    from mpassit_amd import synth
    for order in ("morton", "native"):
        m = synth.icosahedral_mesh(4, order=order)
        assert m.nCells == 10 * 4 ** 4 + 2 and m.nVertices == 20 * 4 ** 4 and m.maxEdges == 6
        nedge = (m.verticesOnCell > 0).sum(1)
        assert (nedge == 5).sum() == 12 and (nedge == 6).sum() == m.nCells - 12
        cxyz, vxyz = mesh_xyz(oracle, m)
        tri, nvalid = oracle.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
        assert nvalid == m.nVertices and (tri >= 0).all()
        # each cell's polygon: counter-clockwise fan areas, together the whole sphere
        total = 0.0
        for c in range(m.nCells):
            v = vxyz[m.verticesOnCell[c, :nedge[c]] - 1]
            for k in range(1, len(v) - 1):
                num = np.dot(v[0], np.cross(v[k] - v[0], v[k + 1] - v[0]))
                assert num > 0
                total += 2 * np.arctan2(num, 1 + v[0] @ v[k] + v[k] @ v[k + 1] + v[k + 1] @ v[0])
        assert abs(total - 4 * np.pi) < 1e-10
    # Morton numbering keeps neighbours close in index space, bisection order does not
    near = synth.icosahedral_mesh(5)
    far = synth.icosahedral_mesh(5, order="native")

    def spread(mm):
        v = mm.verticesOnCell[:, :5].astype(np.int64)
        return np.median(v.max(1) - v.min(1))
    assert spread(near) * 5 < spread(far)


def test_grid_to_grid_bilinear_periodic_monopole(oracle):
    """1PeriDim + MONOPOLE restatement (model_grid.F90:685-694): nothing unmapped on the closed sphere, weights (pole
    share included) sum to one, V at a pole is the zonal mean, row blocks keep only their own caps."""
    from mpassit_amd import target_grid as tg
    t = tg.define_target_grid_params("lat-lon", nx=37, ny=19, stand_lon=0.0, is_regional=False)   # 36 x 18, 10 degrees
    cen = oracle.lonlat_deg_to_xyz(t.lon, t.lat)
    for st, lon, lat in ((1, t.lon_u, t.lat_u), (2, t.lon_v, t.lat_v)):
        idx, w, src0, wp = oracle.grid_bilinear_periodic(t.nx, t.ny, 1, cen, st, oracle.lonlat_deg_to_xyz(lon, lat))
        nxd = lon.shape[1]
        total = w.sum(1)
        total[:nxd] += wp[0]
        total[-nxd:] += wp[1]
        assert (idx[:, 0] >= 0).all() and np.abs(total - 1).max() < 1e-13
        f = np.cos(np.deg2rad(t.lat)) * np.sin(np.deg2rad(t.lon)) + 2.0 + np.sin(np.deg2rad(t.lat))
        out = oracle.apply_grid_periodic(t.nx, idx, w, src0, wp, f.reshape(1, -1), 1).reshape(lon.shape)
        if st == 2:
            assert np.allclose(wp, 1.0, atol=1e-12) and (src0[1] == (t.ny - 1) * t.nx).all() and (src0[0] == 0).all()
            assert np.allclose(out[0], f[0].mean()) and np.allclose(out[-1], f[-1].mean())
        else:
            assert (wp == 0).all() and np.array_equal(out[:, 0], out[:, -1])
        # non-periodic reading of the same grid leaves the seam column / pole rows unmapped
        idx0, _ = oracle.grid_bilinear(t.nx, t.ny, cen, st, oracle.lonlat_deg_to_xyz(lon, lat))
        assert (idx0[:, 0] < 0).sum() == (2 * t.ny if st == 1 else 2 * t.nx)
    # a southern row block: south cap only
    rows = slice(0, 9)
    idx, w, src0, wp = oracle.grid_bilinear_periodic(t.nx, 9, 1 | 4, oracle.lonlat_deg_to_xyz(t.lon[rows], t.lat[rows]), 2,
                                                     oracle.lonlat_deg_to_xyz(t.lon_v[0:10], t.lat_v[0:10]))
    assert np.allclose(wp[0], 1.0, atol=1e-12) and (wp[1] == 0).all() and (idx[-t.nx:, 0] < 0).all()
