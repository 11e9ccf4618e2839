"""Comparison helpers shared by the -m gpu parity tests: every point / matrix entry on which the HIP path and
the oracle differ is EXAMINED (and must be explained by a documented tie or threshold), never just counted."""
import numpy as np

SLIVER = 1e-12       # both implementations drop overlaps below 1e-14 of the destination cell's area
                     # (mpassit_oracle.c orc_conserve / k_store_conserve.hip): entries that exist on one side only
                     # must be slivers at that threshold


def rel_err(a, b):
    scale = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() / scale


def assert_fixed_weights_equal(idx_o, w_o, idx_g, w_g, tol=1e-11, tie_tol=1e-9):
    """[P][k] index/weight sets (-1 = unmapped).  Points with identical index rows: weights within `tol`.
    Every other point must be a shared-edge / shared-vertex tie: as sparse vectors (source id -> weight) the two rows
    agree within `tie_tol`, i.e. the vertices not common to both carry (numerically) zero weight.  Returns the number of
    tie points."""
    assert idx_o.shape == idx_g.shape
    mapped_o, mapped_g = idx_o[:, 0] >= 0, idx_g[:, 0] >= 0
    assert np.array_equal(mapped_o, mapped_g), "mapped masks differ at %s" % np.nonzero(mapped_o != mapped_g)[0][:10]
    same = (idx_o == idx_g).all(axis=1)
    if same.any():
        assert np.abs(w_o[same] - w_g[same]).max() < tol
    for p in np.nonzero(~same)[0]:
        a, b = {}, {}
        for i, w in zip(idx_o[p], w_o[p]):
            a[int(i)] = a.get(int(i), 0.0) + w
        for i, w in zip(idx_g[p], w_g[p]):
            b[int(i)] = b.get(int(i), 0.0) + w
        for k in set(a) | set(b):
            assert abs(a.get(k, 0.0) - b.get(k, 0.0)) < tie_tol, (
                "point %d: oracle %s / %s vs gpu %s / %s is not an edge tie" % (p, idx_o[p], w_o[p], idx_g[p], w_g[p]))
    return int((~same).sum())


def assert_csr_equal(rp_o, col_o, val_o, rp_g, col_g, val_g, n_src, tol=1e-11, sliver=SLIVER):
    """Conservative matrices compared as SETS of (row, col) entries, unconditionally: common entries within `tol`,
    entries present on one side only must be below the sliver threshold.  Returns (n_common, n_only_oracle, n_only_gpu)."""
    P = rp_o.size - 1
    assert rp_g.size - 1 == P
    key_o = np.repeat(np.arange(P, dtype=np.int64), np.diff(rp_o)) * n_src + col_o
    key_g = np.repeat(np.arange(P, dtype=np.int64), np.diff(rp_g)) * n_src + col_g
    assert np.unique(key_o).size == key_o.size and np.unique(key_g).size == key_g.size, "duplicate (row, col) entries"
    so, sg = np.argsort(key_o), np.argsort(key_g)
    key_o, val_o = key_o[so], np.asarray(val_o)[so]
    key_g, val_g = key_g[sg], np.asarray(val_g)[sg]
    common, io, ig = np.intersect1d(key_o, key_g, assume_unique=True, return_indices=True)
    if common.size:
        d = np.abs(val_o[io] - val_g[ig])
        assert d.max() < tol, "entry (row %d, col %d): %r vs %r" % (common[d.argmax()] // n_src, common[d.argmax()] % n_src,
                                                                      val_o[io][d.argmax()], val_g[ig][d.argmax()])
    only_o = np.ones(key_o.size, bool)
    only_o[io] = False
    only_g = np.ones(key_g.size, bool)
    only_g[ig] = False
    for name, keys, vals in (("oracle", key_o[only_o], val_o[only_o]), ("gpu", key_g[only_g], val_g[only_g])):
        if keys.size:
            k = vals.argmax()
            assert vals[k] < sliver, "entry (row %d, col %d) = %r exists only on the %s side and is no sliver" % (
                keys[k] // n_src, keys[k] % n_src, vals[k], name)
    return int(common.size), int(only_o.sum()), int(only_g.sum())


def assert_nearest_equal(idx_o, idx_g, pxyz, cxyz, max_ties=None):
    """Nearest-neighbour indices: identical, except exact ties within rounding of the two sin/cos implementations
    (squared chord distances within 8 ulp); each such point is checked."""
    diff = np.nonzero(idx_g != idx_o)[0]
    for p in diff:
        d_o = np.sum((pxyz[p] - cxyz[idx_o[p]]) ** 2)
        d_g = np.sum((pxyz[p] - cxyz[idx_g[p]]) ** 2)
        assert abs(d_o - d_g) <= 8 * np.finfo(np.float64).eps * d_o, "point %d: %d (d2 %r) vs %d (d2 %r)" % (p, idx_o[p], d_o, idx_g[p], d_g)
    if max_ties is not None:
        assert diff.size <= max_ties
    return int(diff.size)


def conserve_tol(o, g):
    """Tolerance for conservative weights / values on target grid `g`.  A weight is a ratio of two spherical polygon
    areas computed from unit vectors whose coordinates carry 2^-53 rounding.  With the clip planes' normals in difference
    form (oracle and kernel since late round 2; tests/test_weight_goldens.py measures the oracle against 50-digit answers)
    a polygon side sits within ~1e-16 rad of where it should, so the relative error of w = A_overlap / A_dst is
    ~ eps / h with h the THIN dimension of the destination cell (area / longest diagonal: 4.7e-4 rad for a 3-km cell,
    1e-6 for the slivers of a lat-lon grid next to a pole).  64 eps / min(h), floored at 1e-11 and never looser than the
    16 eps / min(A_dst) the direct normals needed -- five to seven orders below the 1e-6 north_star asks for.
    Measured GPU vs oracle on a 3-km grid: 6e-13."""
    c = o.lonlat_deg_to_xyz(g.lon_c, g.lat_c).reshape(g.ny + 1, g.nx + 1, 3)
    d1, d2 = c[1:, 1:] - c[:-1, :-1], c[1:, :-1] - c[:-1, 1:]
    area = 0.5 * np.linalg.norm(np.cross(d1, d2), axis=-1)
    diag = np.maximum(np.linalg.norm(d1, axis=-1), np.linalg.norm(d2, axis=-1))
    thin = area / np.maximum(diag, 1e-300)
    eps = np.finfo(np.float64).eps
    return max(1e-11, min(16 * eps / area.min(), 64 * eps / thin.min()))
