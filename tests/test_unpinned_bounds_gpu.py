"""How much the choices the reference leaves to ESMF -- and ESMF's documentation leaves open -- can move the answer.

Parity with the reference is unpinned at the ESMF boundary (ESMF is absent, the reference holds no fixtures).
Two choices of that kind exist on the path; this file MEASURES what each is worth on the BASELINE meshes and asserts the
measured bound, so that "unpinned" is a stated +-epsilon an integrator with ESMF can confirm, not an unknown:

 (a) node-located bilinear (vorticity, interp.F90:350-366): ESMF splits polygons with more than four sides into triangles
     in an undocumented order; oracle and kernels fan from the first listed vertex.  Rotating every row of verticesOnCell
     gives the same polygons fanned from their 2nd / 3rd vertex -- two other triangulations of every cell;
 (b) bilinear line type: the target point meets the plane of its source triangle along the ray from the sphere's centre
     (default) or along the plane's normal (ESMF_LINETYPE_CART read literally; mpg_tune("bilinear_linetype", 1)).

Fields: synth.analytic_field (a_k + b_k x + c_k y + d_k z + 0.1 sin(5 lon) cos(3 lat)): smooth, O(1) values and
O(1) gradients per radian.  Differences are quoted relative to max |field|."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# measured on MI355X (this file prints the numbers); the asserts allow 1.5x
FAN_SPREAD = {"varres_30k": 1.01e-3, "icosahedral_41k": 3.36e-3}        # max |difference| / max |f|; rms 6.1e-5 / 1.9e-4
LINETYPE_FIELD = {"c1": 2.4e-7, "c2": 3.8e-9, "c4": 4.8e-12}           # regridded smooth field, relative to max |f|
LINETYPE_WEIGHT = {"c1": 2.8e-5, "c2": 1.14e-6, "c4": 1.42e-8}         # max |w_ray - w_normal| over the grid


def _rotate_rows(voc, s):
    out = voc.copy()
    for r in range(voc.shape[0]):
        ne = int((voc[r] > 0).sum())
        out[r, :ne] = np.roll(voc[r, :ne], -s)
    return out


def _meshes():
    from mpassit_amd import synth, target_grid as tg
    g1 = tg.define_target_grid_params("lambert", 241, 161, dx=40000.0, dy=40000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5,
                                      truelat2=38.5, stand_lon=-97.5)
    yield "varres_30k", synth.variable_resolution_mesh(30000), g1
    g2 = tg.define_target_grid_params("lat-lon", 361, 181, stand_lon=0.0, is_regional=False)
    yield "icosahedral_41k", synth.icosahedral_mesh(6), g2


def test_fan_triangulation_spread_of_node_located_bilinear(gpu_lib):
    """Three triangulations of every Voronoi polygon (fans from its 1st, 2nd, 3rd vertex): the regridded vertex field moves
    by O(h^2 |f''|) where a polygon is not a triangle -- the bound an integrator should expect against ESMF's own split."""
    import copy

    from mpassit_amd import regrid as R, synth
    for name, m, g in _meshes():
        grid = R.Grid.from_target(g)
        src = synth.analytic_field(m.latVertex, m.lonVertex, 3)
        res = []
        for s in range(3):
            ms = copy.copy(m)
            ms.verticesOnCell = _rotate_rows(m.verticesOnCell, s)
            mesh = R.Mesh.from_mpas(ms)
            rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR, meshloc=R.MESHLOC_NODE)
            res.append(rh.regrid(src.reshape(-1), nlev=3).copy())
            mapped = rh.weights()[0][:, 0] >= 0
            rh.release()
            mesh.destroy()
        grid.destroy()
        scale = np.abs(res[0]).max()
        spread = max(np.abs(res[s] - res[0]).max() for s in (1, 2)) / scale
        rms = max(np.sqrt(np.mean((res[s] - res[0]) ** 2)) for s in (1, 2)) / scale
        print("node-located bilinear, %s: max spread between fan orders %.3e of max|f| (rms %.3e), %d of %d points mapped" % (
            name, spread, rms, int(mapped.sum()), mapped.size))
        assert 0 < spread < 1.5 * FAN_SPREAD[name], (name, spread)


@pytest.mark.parametrize("cfg", ["c1", "c2", "c4"])
def test_line_type_moves_the_bilinear_answer_by_h_squared(gpu_lib, cfg):
    import torch

    from mpassit_amd import regrid as R, synth, workloads
    m, g, _, desc = workloads.workload({"c1": "c1_65k_global", "c2": "c2_655k_global", "c4": "c4_3m_regional"}[cfg])
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    nlev = 2
    src = torch.as_tensor(synth.analytic_field(m.latCell, m.lonCell, nlev), device="cuda")
    out, wts = [], []
    try:
        for lt in (0, 1):
            gpu_lib.tune("bilinear_linetype", lt)
            rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
            out.append(rh.regrid(src.view(-1), nlev=nlev).cpu().numpy())
            wts.append(rh.weights())
            rh.release()
    finally:
        gpu_lib.tune("bilinear_linetype", 0)
    mesh.destroy()
    grid.destroy()
    (i0, w0), (i1, w1) = wts
    same_tri = (i0 == i1).all(1)
    both = (i0[:, 0] >= 0) & (i1[:, 0] >= 0)
    dw = np.abs(w0[same_tri & both] - w1[same_tri & both]).max()
    df = np.abs(out[0] - out[1])[:, :, both.reshape(out[0].shape[2:])].max() / np.abs(out[0]).max()
    print("line type, %s (%s): weights differ by <= %.3e, the regridded field by <= %.3e of max|f|; %d of %d points change triangle, "
          "%d change mapped / unmapped" % (cfg, desc, dw, df, int((~same_tri & both).sum()), both.size, int(((i0[:, 0] >= 0) != (i1[:, 0] >= 0)).sum())))
    assert dw < 1.5 * LINETYPE_WEIGHT[cfg] and df < 1.5 * LINETYPE_FIELD[cfg], (cfg, dw, df)
    # points that change triangle sit on a shared edge (both triangles give the same value there up to the same O(h^2))
    assert (~same_tri & both).mean() < 1e-3


# ---- round 6: the two remaining shared choices are SELECTABLE (round-5 review, item 7) --------------------------------------
@pytest.mark.parametrize("origin", [0, -1, 1, 3])
def test_node_fan_origin_is_selectable_and_matches_the_oracle(gpu_lib, oracle, origin):
    """mpg_tune("node_fan_origin", o): the fan of every Voronoi polygon starts at its listed vertex number o mod n (-1: the last one --
    what an ear-clipping loop that always cuts the first ear of a convex polygon gives).  Kernels and oracle under the same setting:
    the same triangles win, the same weights; the setting is part of the handle's cache key; the default comes back untouched; and
    origin o on the mesh as given == origin 0 on the mesh with every verticesOnCell row rotated by o (the round-5 spread test's
    construction): one and the same triangulation reached two ways."""
    import copy

    from _parity_helpers import assert_fixed_weights_equal, rel_err
    from conftest import mesh_xyz
    from mpassit_amd import _lib, regrid as R, synth, target_grid as tg
    g = tg.define_target_grid_params("lambert", 121, 81, dx=60000.0, dy=60000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5,
                                     stand_lon=-97.5)
    m = synth.variable_resolution_mesh(12000)
    _, vxyz = mesh_xyz(oracle, m)
    pxyz = oracle.lonlat_deg_to_xyz(g.lon, g.lat)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh0 = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR, meshloc=R.MESHLOC_NODE)         # the default, stored first
    i0, w0 = rh0.weights()
    _lib.tune("node_fan_origin", origin)
    try:
        rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR, meshloc=R.MESHLOC_NODE)
        assert (rh._h.value == rh0._h.value) == (origin == 0)                                   # another setting is another handle
        idx_g, w_g = rh.weights()
    finally:
        _lib.tune("node_fan_origin", 0)
    ftri, nvalid = oracle.fan_triangles(m.verticesOnCell, vxyz, origin=origin)
    assert nvalid > 0
    idx_o, w_o = oracle.bilinear_weights(vxyz, ftri, pxyz)
    n_ties = assert_fixed_weights_equal(idx_o, w_o, idx_g, w_g, tol=1e-11)
    assert n_ties <= 0.002 * idx_o.shape[0]
    src = synth.analytic_field(m.latVertex, m.lonVertex, 3)
    got = rh.regrid(src.reshape(-1), nlev=3).reshape(3, -1)
    assert rel_err(got, oracle.apply_fixed(idx_o, w_o, src, 3)) < 1e-12
    if origin > 0:      # the rotated-rows construction of the spread test: the same field, bit for bit
        ms = copy.copy(m)
        ms.verticesOnCell = _rotate_rows(m.verticesOnCell, origin)
        mesh_r = R.Mesh.from_mpas(ms)
        rh_r = R.regrid_store(mesh_r, grid, R.REGRIDMETHOD_BILINEAR, meshloc=R.MESHLOC_NODE)
        assert np.array_equal(rh_r.regrid(src.reshape(-1), nlev=3).reshape(3, -1), got)
        rh_r.release()
        mesh_r.destroy()
        assert np.abs(got - rh0.regrid(src.reshape(-1), nlev=3).reshape(3, -1)).max() > 0     # ... and it IS another triangulation
    again = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR, meshloc=R.MESHLOC_NODE)         # back at the default: the first handle
    assert again._h.value == rh0._h.value and np.array_equal(again.weights()[1], w0) and np.array_equal(again.weights()[0], i0)
    for r in (rh, rh0, again):
        r.release()
    mesh.destroy()
    grid.destroy()


@pytest.mark.parametrize("exp", [10, 14, 6, 4])
def test_grid_inside_tolerance_is_selectable_and_matches_the_oracle(gpu_lib, oracle, exp):
    """mpg_tune("grid_inside_tol_exp", e): a stagger point counts as inside a quad of CENTER points within 10^-e of its parametric range.
    On a 3-km Lambert grid the EDGE2 points of the first / last column sit ~1e-6 of a cell from the hull of the centres (DESIGN s2): a
    tolerance of 1e-4 maps all of them, 1e-14 maps fewer than the default -- the mapped mask is what an ESMF comparison decides.
    Kernels == oracle under every setting (same mapped mask, same quads, weights to rounding)."""
    from mpassit_amd import _lib, regrid as R, target_grid as tg
    g = tg.define_target_grid_params("lambert", 181, 121, dx=3000.0, dy=3000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5,
                                     stand_lon=-97.5)
    grid = R.Grid.from_target(g)
    cen = oracle.lonlat_deg_to_xyz(g.lon, g.lat)
    masks = {}
    _lib.tune("grid_inside_tol_exp", exp)
    try:
        for so, sg, lon, lat in ((1, R.STAGGERLOC_EDGE1, g.lon_u, g.lat_u), (2, R.STAGGERLOC_EDGE2, g.lon_v, g.lat_v)):
            rh = R.regrid_store_grid(grid, sg)
            gi, gw = rh.weights()
            rh.release()
            oi, ow = oracle.grid_bilinear(g.nx, g.ny, cen, so, oracle.lonlat_deg_to_xyz(lon, lat), tol=10.0 ** -exp)
            mg, mo = gi[:, 0] >= 0, oi[:, 0] >= 0
            # the two Newton solves agree to ~1e-13 in xi / eta: a point within that of the tolerance's edge may fall either way
            assert (mg != mo).sum() <= 4, (exp, so, int((mg != mo).sum()))
            both = mg & mo
            same_quad = (gi[both] == oi[both]).all(axis=1)
            assert same_quad.mean() > 0.999      # (a point ON the border of two quads may take either: the same value, DESIGN s2)
            assert np.abs(gw[both][same_quad] - ow[both][same_quad]).max() < 1e-10
            masks[so] = mg.reshape(lon.shape)
    finally:
        _lib.tune("grid_inside_tol_exp", 10)
    first_last_col = int(masks[2][1:-1, 0].sum() + masks[2][1:-1, -1].sum())     # EDGE2 points of the first / last column, hull rows excluded
    total = 2 * (masks[2].shape[0] - 2)
    print("grid_inside_tol_exp %d: %d of %d first/last-column EDGE2 points mapped" % (exp, first_last_col, total))
    if exp <= 4:
        assert first_last_col == total        # 1e-4 of a cell swallows the 1e-6 bend of the x = const line
    if exp >= 14:
        assert first_last_col < total
    grid.destroy()
