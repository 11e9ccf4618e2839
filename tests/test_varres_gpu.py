"""Variable-resolution global mesh (MPAS 60-3 km style refinement, ratio 8, irregular polygons up to 9+ sides) onto a
Lambert grid that straddles the refined region and the coarse surroundings: every method against the oracle.
Stresses the rasteriser (triangles from << 1 to >> 1 target spacing), the BVH and the polygon clipper."""
import numpy as np
import pytest

from conftest import mesh_xyz

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case(gpu_lib, oracle):
    from mpassit_amd import regrid as R, synth, target_grid as tg
    m = synth.variable_resolution_mesh(30000)
    g = tg.define_target_grid_params("lambert", 241, 161, dx=40000.0, dy=40000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5,
                                     truelat2=38.5, stand_lon=-97.5)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    cxyz, vxyz = mesh_xyz(oracle, m)
    yield dict(m=m, g=g, mesh=mesh, grid=grid, cxyz=cxyz, vxyz=vxyz)
    mesh.destroy()
    grid.destroy()


def test_mesh_is_variable(case):
    m = case["m"]
    ne = (m.verticesOnCell > 0).sum(1)
    assert ne.min() <= 5 and ne.max() >= 8 and m.maxEdges <= 12
    # cell spacing varies by > 4x between the refined cap and the far field
    cx = case["cxyz"]
    c0 = np.array([np.cos(np.deg2rad(38.5)) * np.cos(np.deg2rad(-97.5)), np.cos(np.deg2rad(38.5)) * np.sin(np.deg2rad(-97.5)), np.sin(np.deg2rad(38.5))])
    near = (cx @ c0) > np.cos(np.deg2rad(15))       # refined cap, area 2 pi (1 - cos 15)
    far = (cx @ c0) < 0.0                             # far hemisphere, area 2 pi
    dens_near = near.sum() / (2 * np.pi * (1 - np.cos(np.deg2rad(15))))
    dens_far = far.sum() / (2 * np.pi)
    assert dens_near / dens_far > 16                  # spacing ratio > 4


def test_all_methods_match_oracle(case, oracle):
    from mpassit_amd import regrid as R, synth
    m, g = case["m"], case["g"]
    pxyz = oracle.lonlat_deg_to_xyz(g.lon, g.lat)
    tri, _ = oracle.dual_triangles(m.verticesOnCell, m.nVertices, case["cxyz"])
    assert np.array_equal(case["mesh"].triangles(), tri)
    src = synth.analytic_field(m.latCell, m.lonCell, 4)
    # bilinear
    idx_o, w_o = oracle.bilinear_weights(case["cxyz"], tri, pxyz)
    rh = R.regrid_store(case["mesh"], case["grid"], R.REGRIDMETHOD_BILINEAR)
    idx_g, w_g = rh.weights()
    assert (idx_g >= 0).all() and np.array_equal(idx_o[:, 0] >= 0, idx_g[:, 0] >= 0)
    same = (idx_o == idx_g).all(1)
    assert same.mean() > 0.999 and np.abs(w_o[same] - w_g[same]).max() < 1e-10
    got = rh.regrid(src, nlev=4).reshape(4, -1)
    want = oracle.apply_fixed(idx_o, w_o, src, 4)
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-11
    rh.release()
    # nearest
    rh = R.regrid_store(case["mesh"], case["grid"], R.REGRIDMETHOD_NEAREST_STOD)
    assert np.array_equal(rh.weights()[0][:, 0], oracle.nearest(case["cxyz"], pxyz))
    rh.release()
    # conservative
    pc = oracle.lonlat_deg_to_xyz(g.lon_c, g.lat_c)
    rp_o, col_o, val_o = oracle.conserve(m.verticesOnCell, case["vxyz"], g.nx, g.ny, pc)
    rh = R.regrid_store(case["mesh"], case["grid"], R.REGRIDMETHOD_CONSERVE)
    rp_g, col_g, val_g = rh.csr()
    P = g.nx * g.ny
    rows = np.repeat(np.arange(P), np.diff(rp_g))
    assert np.abs(np.bincount(rows, weights=val_g, minlength=P) - 1).max() < 1e-9
    snow = synth.snow_field(m.latCell, m.lonCell)
    got = rh.regrid(snow, nlev=1).reshape(-1)
    want = oracle.apply_csr(rp_o, col_o, val_o, snow, 1)[0]
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-10
    rh.release()
    # node-located
    ftri, _ = oracle.fan_triangles(m.verticesOnCell, case["vxyz"])
    ix, wx = oracle.bilinear_weights(case["vxyz"], ftri, pxyz)
    rh = R.regrid_store(case["mesh"], case["grid"], R.REGRIDMETHOD_BILINEAR, meshloc=R.MESHLOC_NODE)
    vort = np.random.default_rng(2).standard_normal((2, m.nVertices))
    got = rh.regrid(vort, nlev=2).reshape(2, -1)
    want = oracle.apply_fixed(ix, wx, vort, 2)
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-11
    rh.release()
