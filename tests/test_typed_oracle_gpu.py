"""Direct oracle parity of the fused ingest/egress Regrid (mpg_regrid_typed_dev) and of the staged kernels: every
combination of source layout (cell-fast / MPAS file order), element types (float32 / float64 on either side) and kernel
family (library default, forced lane-/row-gather, forced LDS-staged) against the ORACLE's
chain -- widen at read (input_data.F90:630-655), float64 weights from the oracle's own search, float64 apply,
post-op, narrow at write (write_data.F90:779,1343,1418) -- not against another GPU kernel."""
import numpy as np
import pytest

from _parity_helpers import rel_err
from conftest import mesh_xyz

pytestmark = pytest.mark.gpu

NLEV = 55                      # the level count of the BASELINE configs (odd: the row kernels' last lane shifts)
KERNELS = [                    # (a3_staged, lf_variant): which kernel family serves the 3-point handle
    (-1, -1),                  # library default (per-handle choice)
    (-2, 2),                   # lane gather (cell-fast) / row gather on grid-row tiles (level-fast)
    (0, 0),                    # LDS-staged 64x8 tiles / row gather on linear aligned tiles
    (1, 1),                    # LDS-staged 64x16 tiles on 256 threads / LDS-staged in 16-level chunks
    (2, 1),                    # LDS-staged 64x16 tiles on 512 threads
]


def _same_as_oracle_f32(got32, ref32, what):
    """equal to the oracle's narrowing except where the two float64 values (<= 1e-12 apart) straddle a float32 rounding
    boundary -> at most one float32 ulp, on a vanishing fraction of the points"""
    ne = got32 != ref32
    assert ne.mean() < 1e-5, (what, ne.mean())
    if ne.any():
        assert (np.abs(got32[ne].astype(np.float64) - ref32[ne]) <= np.spacing(np.abs(ref32[ne])).astype(np.float64)).all(), what


def _cases():
    from mpassit_amd import synth, target_grid as T, workloads
    m, g, _, _ = workloads.workload("tiny")                                  # regional lattice, 1.6 cells per point, unmapped rim
    yield "regional", m, g
    g2 = T.define_target_grid_params("lat-lon", nx=331, ny=167, stand_lon=0.0, is_regional=False)
    yield "global_reuse5", synth.icosahedral_mesh(5), g2                     # 5 target points per cell: staged kernels by default


@pytest.mark.parametrize("case", ["regional", "global_reuse5"])
def test_typed_and_staged_kernels_against_oracle(gpu_lib, oracle, case):
    import torch
    from mpassit_amd import regrid as R
    name, m, g = next(c for c in _cases() if c[0] == case)
    cxyz, _ = mesh_xyz(oracle, m)
    tri, _ = oracle.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    idx_o, w_o = oracle.bilinear_weights(cxyz, tri, oracle.lonlat_deg_to_xyz(g.lon, g.lat))
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    rng = np.random.default_rng(17)
    nf = 2
    s64 = rng.normal(size=(nf, NLEV, m.nCells)) * 30.0 + 280.0
    s32 = s64.astype(np.float32)
    scale, offset = 9.81, -300.0
    # the oracle's chain, once per source type
    want = {}
    for sname, s in (("f64", s64), ("f32", s32.astype(np.float64))):
        r = np.stack([oracle.apply_fixed(idx_o, w_o, s[f], NLEV) for f in range(nf)])      # [nf][nlev][P] float64
        want[sname] = r
    checked = 0
    try:
        for staged, lfv in KERNELS:
            gpu_lib.tune("a3_staged", staged)
            gpu_lib.tune("lf_variant", lfv)
            for sname, s in (("f64", s64), ("f32", s32)):
                for layout in (R.LAYOUT_CELL_FAST, R.LAYOUT_LEV_FAST):
                    src = s if layout == R.LAYOUT_CELL_FAST else np.ascontiguousarray(s.transpose(0, 2, 1))
                    src_t = torch.as_tensor(src, device="cuda").view(-1)
                    ref = want[sname]
                    # float64 out, identity epilogue: the plain Regrid of the reference
                    got = rh.regrid_typed(src_t, nlev=NLEV, nfields=nf, layout=layout, out_dtype=torch.float64).cpu().numpy().reshape(ref.shape)
                    assert rel_err(got, ref) < 1e-12, (name, staged, lfv, sname, layout)
                    # float32 out with the writer's affine post-op: equal to the oracle's narrowing except where the two
                    # float64 values (<= 1e-12 apart) straddle a float32 rounding boundary -> at most one float32 ulp,
                    # on a vanishing fraction of the points
                    ref32 = (ref * scale + offset).astype(np.float32)
                    got32 = rh.regrid_typed(src_t, nlev=NLEV, nfields=nf, layout=layout, out_dtype=torch.float32, scale=scale,
                                            offset=offset).cpu().numpy().reshape(ref.shape)
                    _same_as_oracle_f32(got32, ref32, (name, staged, lfv, sname, layout))
                    # the same with both sides BIG-ENDIAN, as the Fortran driver's file flow runs it: the source holds the bytes
                    # of a NetCDF classic variable (input_data.F90:630 reads them through nf90_get_var), the result the bytes
                    # nf90_put_var would store (write_data.F90:1339-1475) -- against the oracle's chain, and bit for bit against
                    # the host-order call
                    src_be_t = torch.as_tensor(src.byteswap().view(src.dtype), device="cuda").view(-1)
                    be32 = rh.regrid_typed(src_be_t, nlev=NLEV, nfields=nf, layout=layout, out_dtype=torch.float32, scale=scale, offset=offset,
                                           src_be=True, dst_be=True).cpu().numpy().reshape(ref.shape)
                    be32 = be32.view(">f4").astype(np.float32)
                    _same_as_oracle_f32(be32, ref32, (name, staged, lfv, sname, layout, "be"))
                    assert np.array_equal(be32.view(np.uint32), got32.view(np.uint32)), (name, staged, lfv, sname, layout, "be != le")
                    # mixed: big-endian source, float64 host-order result (what the driver keeps for rotation / P_TOP / Z_C)
                    be64 = rh.regrid_typed(src_be_t, nlev=NLEV, nfields=nf, layout=layout, out_dtype=torch.float64, src_be=True).cpu().numpy()
                    assert np.array_equal(be64.reshape(ref.shape), got), (name, staged, lfv, sname, layout, "be -> f64")
                    checked += 1
            # the plain float64 entry point (mpg_regrid_dev) in both layouts under the same kernel choice
            for layout in (R.LAYOUT_CELL_FAST, R.LAYOUT_LEV_FAST):
                src = s64 if layout == R.LAYOUT_CELL_FAST else np.ascontiguousarray(s64.transpose(0, 2, 1))
                got = rh.regrid(torch.as_tensor(src, device="cuda").view(-1), nlev=NLEV, nfields=nf, layout=layout).cpu().numpy()
                assert rel_err(got.reshape(want["f64"].shape), want["f64"]) < 1e-12, (name, staged, lfv, layout)
    finally:
        gpu_lib.tune("lf_variant", -1)
        gpu_lib.tune("a3_staged", -1)
    assert checked == len(KERNELS) * 4
    rh.release()
    mesh.destroy()
    grid.destroy()
