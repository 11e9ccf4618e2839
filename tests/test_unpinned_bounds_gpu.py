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
