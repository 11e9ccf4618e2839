"""BASELINE configs[3] (3.0 M cells x 55 levels -> 1800x1060) at full size: size-independent properties (partition of
unity, exact reproduction of constants, linearity, closed-form answer for Cartesian-linear fields, nearest-neighbour
optimality on a sample and value-set preservation, conservative row sums / non-negativity, equality of the two source
layouts, determinism of repeated Stores) AND the oracle itself at full size (test_c4_full_size_oracle_parity: all three
weight sets entry by entry, one whole 55-level field)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c4(gpu_lib):
    import torch
    from mpassit_amd import regrid as R, workloads
    m, g, nlev, _ = workloads.workload("c4_3m_regional")
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    yield dict(m=m, g=g, mesh=mesh, grid=grid, nlev=nlev, torch=torch)
    mesh.destroy()
    grid.destroy()


def _xyz(lat, lon):
    cl = np.cos(lat)
    return np.stack([cl * np.cos(lon), cl * np.sin(lon), np.sin(lat)], axis=-1)


def test_c4_bilinear_properties(c4):
    from mpassit_amd import regrid as R
    torch, m, g = c4["torch"], c4["m"], c4["g"]
    rh = R.regrid_store(c4["mesh"], c4["grid"], R.REGRIDMETHOD_BILINEAR)
    assert rh.n_dst == 1800 * 1060 and rh.n_src == m.nCells
    idx, w = rh.weights()
    mapped = idx[:, 0] >= 0
    assert mapped.all()                                   # the 3 M-cell mesh covers the whole Lambert domain
    assert np.abs(w.sum(1) - 1).max() < 1e-13 and w.min() > -1e-10
    # closed form: f = a.X  ->  t*(a.P), t = det[A,B,C]/S, on every one of the 1.9 M points
    cx = _xyz(m.latCell, np.where(m.lonCell > np.pi, m.lonCell - 2 * np.pi, m.lonCell))
    px = _xyz(np.deg2rad(g.lat).ravel(), np.deg2rad(g.lon).ravel())
    a = np.array([0.3, -0.7, 0.5])
    src = torch.as_tensor(np.stack([cx @ a, np.full(m.nCells, 7.25)]), device="cuda")       # 2 "levels"
    out = rh.regrid(src.view(-1), nlev=2).cpu().numpy().reshape(2, -1)
    A, B, C = cx[idx[:, 0]], cx[idx[:, 1]], cx[idx[:, 2]]
    n = np.cross(B - A, C - A)                     # difference form: well conditioned for 3-km triangles
    t = np.einsum("ij,ij->i", A, n) / np.einsum("ij,ij->i", n, px)
    assert np.abs(out[0] - t * (px @ a)).max() < 1e-12
    assert np.abs(out[1] - 7.25).max() < 1e-13           # constants are reproduced (sum w = 1)
    # linearity over a 55-level bundle + both source layouts agree
    L = c4["nlev"]
    x = torch.randn((2, L, m.nCells), dtype=torch.float64, device="cuda")
    y = rh.regrid(x.view(-1), nlev=L, nfields=2)
    z = rh.regrid((2.0 * x[0] - 3.0 * x[1]).contiguous().view(-1), nlev=L)
    assert float((z[0] - (2.0 * y[0] - 3.0 * y[1])).abs().max()) < 1e-12
    ylf = rh.regrid(x.permute(0, 2, 1).contiguous().view(-1), nlev=L, nfields=2, layout=R.LAYOUT_LEV_FAST)
    assert float((ylf - y).abs().max()) < 1e-13
    # min/max principle: convex weights (up to the 1e-10 edge tolerance)
    assert float(y.max()) <= float(x.max()) + 1e-9 and float(y.min()) >= float(x.min()) - 1e-9
    print("store_ms bilinear C4:", rh.store_ms)
    rh.release()
    # determinism: a fresh Store gives the identical handle contents
    rh2 = R.regrid_store(c4["mesh"], c4["grid"], R.REGRIDMETHOD_BILINEAR)
    idx2, w2 = rh2.weights()
    assert np.array_equal(idx, idx2) and np.array_equal(w, w2)
    rh2.release()


def test_c4_nearest_properties(c4):
    from mpassit_amd import regrid as R, synth
    torch, m, g = c4["torch"], c4["m"], c4["g"]
    rh = R.regrid_store(c4["mesh"], c4["grid"], R.REGRIDMETHOD_NEAREST_STOD)
    idx, _ = rh.weights()
    idx = idx[:, 0]
    assert idx.min() >= 0 and idx.max() < m.nCells
    cx = _xyz(m.latCell, m.lonCell)
    px = _xyz(np.deg2rad(g.lat).ravel(), np.deg2rad(g.lon).ravel())
    rng = np.random.default_rng(0)
    sample = rng.choice(px.shape[0], 40, replace=False)
    for p in sample:                                       # brute force over all 3 M cells
        d = ((cx - px[p]) ** 2).sum(1)
        assert d[idx[p]] <= d.min() * (1 + 1e-12)
    cat = synth.category_field(m.nCells, nlev=2)
    out = rh.regrid(torch.as_tensor(cat, device="cuda").view(-1), nlev=2).cpu().numpy().reshape(2, -1)
    assert np.array_equal(out, cat[:, idx])                # bit-exact copy, integer categories preserved
    print("store_ms nearest C4:", rh.store_ms)
    rh.release()


def test_c4_conservative_properties(c4):
    from mpassit_amd import regrid as R
    torch, m, g = c4["torch"], c4["m"], c4["g"]
    rh = R.regrid_store(c4["mesh"], c4["grid"], R.REGRIDMETHOD_CONSERVE)
    rp, col, val = rh.csr()
    P = g.nx * g.ny
    assert val.min() > 0 and col.min() >= 0 and col.max() < m.nCells
    rows = np.repeat(np.arange(P), np.diff(rp))
    rowsum = np.bincount(rows, weights=val, minlength=P)
    assert np.abs(rowsum - 1).max() < 1e-9                 # every destination cell fully covered by the mesh
    assert (np.diff(rp) >= 1).all() and np.diff(rp).max() < 32
    for p in np.random.default_rng(1).choice(P, 1000, replace=False):   # rows sorted by source id, no duplicates
        c = col[rp[p]:rp[p + 1]]
        assert (np.diff(c) > 0).all()
    const = torch.full((m.nCells,), 4.5, dtype=torch.float64, device="cuda")
    out = rh.regrid(const, nlev=1).cpu().numpy().ravel()
    assert np.abs(out - 4.5).max() < 1e-8
    print("store_ms conserve C4:", rh.store_ms, "nnz/row", len(col) / P)
    rh.release()


def test_c4_full_size_oracle_parity(c4, oracle):
    """configs[3] against the ORACLE at full size (its hashed searches and threaded loops finish 3.0 M cells x 1.9 M
    points in ~20 s): the three weight sets entry by entry -- every differing entry examined -- and one whole 55-level
    field through the default Regrid kernel."""
    from _parity_helpers import assert_csr_equal, assert_fixed_weights_equal, assert_nearest_equal, conserve_tol, rel_err
    from conftest import mesh_xyz
    from mpassit_amd import regrid as R, synth
    torch, m, g, L = c4["torch"], c4["m"], c4["g"], c4["nlev"]
    o = oracle
    cxyz, vxyz = mesh_xyz(o, m)
    tri, _ = o.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    assert np.array_equal(c4["mesh"].triangles(), tri)
    pxyz = o.lonlat_deg_to_xyz(g.lon, g.lat)
    idx_o, w_o = o.bilinear_weights(cxyz, tri, pxyz)
    rh = R.regrid_store(c4["mesh"], c4["grid"], R.REGRIDMETHOD_BILINEAR)
    idx_g, w_g = rh.weights()
    n_ties = assert_fixed_weights_equal(idx_o, w_o, idx_g, w_g, tol=1e-10)   # 3-km triangles: weights are ratios of 1e-7 areas
    assert n_ties <= 1e-4 * idx_o.shape[0]
    src = synth.analytic_field(m.latCell, m.lonCell, L)
    got = rh.regrid(torch.as_tensor(src, device="cuda").view(-1), nlev=L).cpu().numpy().reshape(L, -1)
    assert rel_err(got, o.apply_fixed(idx_o, w_o, src, L)) < 1e-11 < 1e-6
    got_lf = rh.regrid(torch.as_tensor(np.ascontiguousarray(src.T), device="cuda").view(-1), nlev=L, layout=R.LAYOUT_LEV_FAST)
    assert np.array_equal(got_lf.cpu().numpy().reshape(L, -1), got)
    rh.release()
    rh = R.regrid_store(c4["mesh"], c4["grid"], R.REGRIDMETHOD_NEAREST_STOD)
    assert_nearest_equal(o.nearest(cxyz, pxyz), rh.weights()[0][:, 0], pxyz, cxyz, max_ties=4)
    rh.release()
    rh = R.regrid_store(c4["mesh"], c4["grid"], R.REGRIDMETHOD_CONSERVE)
    rp_o, col_o, val_o = o.conserve(m.verticesOnCell, vxyz, g.nx, g.ny, o.lonlat_deg_to_xyz(g.lon_c, g.lat_c))
    rp_g, col_g, val_g = rh.csr()
    tol = conserve_tol(o, g)
    n_common, only_o, only_g = assert_csr_equal(rp_o, col_o, val_o, rp_g, col_g, val_g, m.nCells, tol=tol)
    assert n_common > 0.999 * max(col_o.size, col_g.size)
    snow = synth.snow_field(m.latCell, m.lonCell)
    assert rel_err(rh.regrid(snow, nlev=1).reshape(1, -1), o.apply_csr(rp_o, col_o, val_o, snow, 1)) < tol < 1e-6
    rh.release()


def test_c4_morton_numbering_same_answer(c4):
    """configs[3] with a realistic cell numbering (workload c4_3m_morton: the same 3.0 M cells renumbered along a Morton
    curve instead of row by row): the geometry is unchanged, so every method must give the same answer on the permuted
    fields -- nearest bit for bit, bilinear to rounding (the dual triangles list their corners in another order) -- and
    the library's tile statistics must report the poorer line locality of the numbering."""
    from mpassit_amd import regrid as R, synth
    torch, m, g, L = c4["torch"], c4["m"], c4["g"], c4["nlev"]
    lon = np.where(m.lonCell > np.pi, m.lonCell - 2.0 * np.pi, m.lonCell)
    mm = synth.morton_cells(m)
    # recover the permutation (new cell i = old cell perm[i]) from the coordinates
    key_old = np.lexsort((m.lonCell, m.latCell))
    key_new = np.lexsort((mm.lonCell, mm.latCell))
    perm = np.empty(m.nCells, np.int64)
    perm[key_new] = key_old
    assert np.array_equal(m.latCell[perm], mm.latCell) and np.array_equal(m.lonCell[perm], mm.lonCell)
    assert np.abs(np.diff(perm)).mean() > 100            # not row-banded any more
    mesh_m = R.Mesh.from_mpas(mm)
    src = synth.analytic_field(m.latCell, m.lonCell, L)
    src_m = np.ascontiguousarray(src[:, perm])
    stats = {}
    for name, meshh, s in (("rows", c4["mesh"], src), ("morton", mesh_m, src_m)):
        rh = R.regrid_store(meshh, c4["grid"], R.REGRIDMETHOD_BILINEAR)
        out = rh.regrid(torch.as_tensor(s, device="cuda").view(-1), nlev=L)
        ts = rh.tile_stats()                             # of the cell-fast kernel's tile lists
        out_lf = rh.regrid(torch.as_tensor(np.ascontiguousarray(s.T), device="cuda").view(-1), nlev=L, layout=R.LAYOUT_LEV_FAST)
        assert torch.equal(out, out_lf)
        stats[name] = (out.cpu().numpy(), ts, rh.kernel_choice())
        rh.release()
    a, b = stats["rows"][0], stats["morton"][0]
    assert np.abs(a - b).max() < 1e-12 * np.abs(a).max()
    (tx, ty, reuse_r, fill_r), (_, _, reuse_m, fill_m) = stats["rows"][1], stats["morton"][1]
    assert abs(reuse_r - reuse_m) < 0.05 * reuse_r       # the same cells per tile, only their ids differ
    assert fill_r > 0.7 and fill_m < 0.9 * fill_r        # Morton blocks: a tile's cells are shorter runs of consecutive ids
    print("tile lists %dx%d: reuse %.2f; line fill row-numbered %.2f, Morton-numbered %.2f" % (tx, ty, reuse_r, fill_r, fill_m))
    rn_r = R.regrid_store(c4["mesh"], c4["grid"], R.REGRIDMETHOD_NEAREST_STOD)
    rn_m = R.regrid_store(mesh_m, c4["grid"], R.REGRIDMETHOD_NEAREST_STOD)
    assert np.array_equal(perm[rn_m.weights()[0][:, 0]], rn_r.weights()[0][:, 0])
    rn_r.release()
    rn_m.release()
    mesh_m.destroy()


def test_c4_row_blocks_on_windowed_meshes_equal_the_whole_mesh(c4):
    """Configuration 4 as its 8-GPU run sees it: a rank's mesh is cut to its block of target rows (mpg_mesh_create_window) --
    380 k of the 3.0 M cells resident for an eighth of the rows -- and every Store on it gives the weights the whole mesh gives,
    bit for bit: three of the eight blocks (first, a middle one, last), all three methods, on the one card."""
    from mpassit_amd import dist, regrid as R
    m, g = c4["m"], c4["g"]
    for rank in (0, 3, 7):
        j0, j1 = dist.row_block(g.ny, 8, rank)
        grid = R.Grid.from_target(g, rows=(j0, j1))
        cut = R.Mesh.from_mpas(m, window_grid=grid)
        c0, cn, v0, vn, margin = cut.window_info()
        assert 0.10 * m.nCells < cn < 0.16 * m.nCells and 0.10 * m.nVertices < vn < 0.16 * m.nVertices, (rank, cn, vn)
        for method in (R.REGRIDMETHOD_BILINEAR, R.REGRIDMETHOD_NEAREST_STOD, R.REGRIDMETHOD_CONSERVE):
            a, b = R.regrid_store(c4["mesh"], grid, method), R.regrid_store(cut, grid, method)
            wa, wb = (a.csr(), b.csr()) if method == R.REGRIDMETHOD_CONSERVE else (a.weights(), b.weights())
            assert a.n_src == b.n_src == m.nCells
            for x, y in zip(wa, wb):
                assert np.array_equal(x, y), (rank, method)
            a.release()
            b.release()
        cut.destroy()
        grid.destroy()


@pytest.mark.parametrize("workload", ["x_c4_polar", "x_c4_mercator"])
def test_c4_sized_polar_stereographic_and_mercator_grids(gpu_lib, oracle, workload):
    """Round 5: configuration 4's sizes (3.0 M cells, 1800 x 1060 mass points at 3 km) on the other two projections of the namelist
    (program_setup.F90:174-182) -- a polar stereographic grid with the north pole inside it, a Mercator grid across the date line.
    The three Stores go through the grid's index space (inverses of module_map_utils.F90:718-760 / :1320-1341) and are compared with
    the ORACLE entry by entry; their times are printed beside the Lambert grid's (0.31 / 0.6 / 2.7 ms at the end of round 4)."""
    from test_configs_gpu import OracleSide, check_stores
    from mpassit_amd import regrid as R, workloads
    m, g, _, _ = workloads.workload(workload)
    osd = OracleSide(oracle, m, g)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_proj(g)
    res = check_stores(R, mesh, grid, osd, ("bilinear", "nearest", "conserve"))
    assert res["mapped"].all()
    ms = {}
    for rep in range(2):            # fresh objects: nothing from the handle cache; the second round is the warm figure
        mesh2, grid2 = R.Mesh.from_mpas(m), R.Grid.from_proj(g)
        for name, code in (("bilinear", R.REGRIDMETHOD_BILINEAR), ("nearest", R.REGRIDMETHOD_NEAREST_STOD), ("conserve", R.REGRIDMETHOD_CONSERVE)):
            rh = R.regrid_store(mesh2, grid2, code)
            assert rh.store_path >= 1, (name, rh.store_stats)          # through the index space, not the pyramid / the tree alone
            ms[name] = rh.store_ms
            rh.release()
        mesh2.destroy()
        grid2.destroy()
    print("store_ms %s:" % workload, {k: round(v, 3) for k, v in ms.items()})
    assert ms["bilinear"] < 1.0 and ms["nearest"] < 1.5 and ms["conserve"] < 5.0, ms       # the pyramid walks took 1.8 / 3.0 / 4.4 ms
    mesh.destroy()
    grid.destroy()
