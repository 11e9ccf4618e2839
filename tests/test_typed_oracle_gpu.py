"""Direct oracle parity of the fused ingest/egress Regrid (mpg_regrid_typed_dev) and of the staged kernels: every
combination of source layout (cell-fast / MPAS file order), element types (float32 / float64 on either side) and kernel
family (library default, forced lane-/row-gather incl. k_apply3_lf_f32x2, forced LDS-staged) against the ORACLE's
chain -- widen at read (input_data.F90:630-655), float64 weights from the oracle's own search, float64 apply,
post-op, narrow at write (write_data.F90:779,1343,1418) -- not against another GPU kernel."""
import numpy as np
import pytest

from _parity_helpers import rel_err
from conftest import mesh_xyz

pytestmark = pytest.mark.gpu

NLEV = 55                      # the level count of the BASELINE configs (odd: the f32x2 kernel's last lane shifts)
KERNELS = [                    # (a3_staged, lf_variant): which kernel family serves the 3-point handle
    (-1, -1),                  # library default (per-handle choice)
    (-2, 4),                   # lane-gather (cell-fast) / row-gather (level-fast; float32 rows -> k_apply3_lf_f32x2)
    (1, 100),                  # LDS-staged cell-fast / LDS-staged level-fast, forced
    (13, 106),                 # the tile shapes bench.py's default run ends up with on C4 / the pipelined staged variant
    (-1, 300), (-1, 301), (-1, 302),   # rows-resident level-fast kernel k_apply3_lfs, tiles 64x1 / 64x2 / 64x4
    (-2, 401), (-2, 410), (-2, 413), (-2, 417),   # float32 row gather: 64x1 tiles at 8 waves / SIMD; linear tiles, 1 / all fields per WG, unroll 4
]


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
                    ne = got32 != ref32
                    assert ne.mean() < 1e-5, (name, staged, lfv, sname, layout, ne.mean())
                    if ne.any():
                        assert (np.abs(got32[ne].astype(np.float64) - ref32[ne]) <= np.spacing(np.abs(ref32[ne])).astype(np.float64)).all()
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
