"""Source window of a mesh (mpg_mesh_set_source_window): a host that holds only the contiguous range of cells its target
rows reference -- the Fortran driver's images read just that range of every MPAS variable.  The windowed Regrid of the
slab must equal the whole-mesh Regrid bit for bit, for every method, handles made before and after the window was set."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_windowed_regrid_equals_whole_mesh_regrid(gpu_lib, regional_case):
    from mpassit_amd import regrid as R
    from mpassit_amd._lib import MpgError
    m, g = regional_case
    rows = (20, 55)                                               # a row block of the grid: references a band of the mesh
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g, rows=rows)
    nlev = 5
    rng = np.random.default_rng(3)
    src = rng.normal(size=(nlev, m.nCells))
    src_lf = np.ascontiguousarray(src.T)
    methods = (R.REGRIDMETHOD_BILINEAR, R.REGRIDMETHOD_NEAREST_STOD, R.REGRIDMETHOD_CONSERVE)
    hs = [R.regrid_store(mesh, grid, md) for md in methods[:2]]          # two exist before the window is set
    want = [h.regrid(src.reshape(-1), nlev=nlev) for h in hs]
    rngs = [h.source_range() for h in hs]
    lo, hi = min(a for a, _ in rngs), max(b for _, b in rngs)
    assert 0 < lo < hi < m.nCells                                 # the band is a proper part of the mesh
    ids = hs[0].unique_sources()
    assert rngs[0] == (int(ids.min()), int(ids.max()) + 1)
    hs[1].release()                                               # parked in the cache: moves with the window all the same
    # a window that cuts into what a handle references is refused and changes nothing
    with pytest.raises(MpgError):
        mesh.set_source_window(lo + 1, hi - lo - 1)
    assert np.array_equal(hs[0].regrid(src.reshape(-1), nlev=nlev), want[0])
    hc = R.regrid_store(mesh, grid, R.REGRIDMETHOD_CONSERVE)      # conservative: its own (wider) range
    a, b = hc.source_range()
    lo, hi = min(lo, a), max(hi, b)
    want.append(hc.regrid(src.reshape(-1), nlev=nlev))
    mesh.set_source_window(lo, hi - lo)
    hs[1] = R.regrid_store(mesh, grid, methods[1])                # back from the cache, already windowed
    hs.append(hc)
    for h in hs:
        h._refresh()
        assert h.n_src == hi - lo
        assert h.source_range()[0] >= lo and h.source_range()[1] <= hi      # still reported as global ids
    # ids at the boundary stay GLOBAL whatever the window is; the in-place re-indexing verbs (the halo route) and a window
    # are alternatives and refuse each other instead of mixing window-relative with global ids
    assert np.array_equal(hs[0].unique_sources(), ids)
    with pytest.raises(MpgError, match="source window"):
        hs[0].rebase(lo, hi - lo)
    with pytest.raises(MpgError, match="source window"):
        hs[0].localize()
    slab, slab_lf = np.ascontiguousarray(src[:, lo:hi]), np.ascontiguousarray(src_lf[lo:hi])
    for h, w in zip(hs, want):
        assert np.array_equal(h.regrid(slab.reshape(-1), nlev=nlev), w)
        assert np.array_equal(h.regrid(slab_lf.reshape(-1), nlev=nlev, layout=R.LAYOUT_LEV_FAST), w)
    # a handle stored AFTER the window was set (another stagger) is windowed from the start
    hu = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD, staggerloc=R.STAGGERLOC_EDGE1)
    assert hu.n_src == hi - lo
    got = hu.regrid(slab.reshape(-1), nlev=nlev)
    mesh.set_source_window(0, m.nCells)                           # back to the whole mesh
    hu._refresh()
    assert hu.n_src == m.nCells and np.array_equal(hu.regrid(src.reshape(-1), nlev=nlev), got)
    for h, w in zip(hs, want):
        h._refresh()
        assert np.array_equal(h.regrid(src.reshape(-1), nlev=nlev), w)
    for h in hs + [hu]:
        h.release()
    mesh.destroy()
    grid.destroy()
