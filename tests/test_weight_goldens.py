"""The oracle's two weight formulas against tests/golden/weights_hp.json: 160 (triangle, point) pairs and 30 (polygon, 4 x 4
grid) cases from 3-km to 500-km cells, evaluated by tests/golden/make_weight_goldens.py along a different route at 50 digits
(3 x 3 solve instead of the determinant closed form; Girard's angle excess instead of the triangle fan).  Inputs are the
stored float64 values exactly, so what is compared is the double-precision arithmetic of the oracle alone; the kernels are
tied to the oracle by the parity tests (tests/test_parity_gpu.py, tests/test_configs_gpu.py).  Not a pin on ESMF."""
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "weights_hp.json")))


def test_bilinear_triangle_weights(oracle):
    worst = {}
    for c in GOLD["bilinear"]:
        tri = np.array(c["tri"])
        idx, w = oracle.bilinear_weights(tri, np.array([[0, 1, 2]], np.int32), np.array([c["p"]]))
        assert sorted(idx[0].tolist()) == [0, 1, 2], c
        got = np.empty(3)
        got[idx[0]] = w[0]                                   # weight of vertex k, whatever order the triangle is stored in
        err = np.abs(got - np.array(c["w"])).max()
        worst[c["scale_km"]] = max(worst.get(c["scale_km"], 0.0), err)
        assert abs(got.sum() - 1) < 4e-16
    # the difference form keeps km-scale triangles as accurate as continental ones
    assert max(worst.values()) < 2e-13, worst


def test_conservative_overlap_weights(oracle):
    worst = {}
    for c in GOLD["conserve"]:
        src, cor = np.array(c["src"]), np.array(c["corners"])
        nv = src.shape[0]
        voc = np.zeros((1, 8), np.int32)
        voc[0, :nv] = np.arange(1, nv + 1)
        rp, col, val = oracle.conserve(voc, src, 4, 4, cor)
        got = np.zeros(16)
        for p in range(16):
            if rp[p + 1] > rp[p]:
                assert rp[p + 1] - rp[p] == 1 and col[rp[p]] == 0
                got[p] = val[rp[p]]
        want = np.array(c["w"])
        sliver = (want < 1e-12) | (got == 0.0) & (want < 1e-11)      # the code drops overlaps below 1e-14 of the cell
        err = np.abs(got - want)[~sliver].max(initial=0.0)
        key = (c["scale_km"], c["ratio"])
        worst[key] = max(worst.get(key, 0.0), err)
        assert (got[sliver] < 1e-11).all()
        assert want.max() > 0.01                                     # the case does overlap
    # unit vectors carry 1e-16 of absolute position: a cell of angular size h has its area to ~1e-16 / h relative
    for (scale_km, ratio), e in worst.items():
        h = scale_km * ratio / 6370.0
        assert e < 64 * 2.2e-16 / h, (scale_km, ratio, e, worst)


def test_quad_bilinear_weights(oracle):
    """Grid -> grid bilinear (CENTER -> EDGE, the destaggering of U and V): four weights per point on a 2 x 2 CENTER grid."""
    worst = {}
    for c in GOLD["quad"]:
        idx, w = oracle.grid_bilinear(2, 2, np.array(c["centres"]), 1, np.array(c["pts"]))
        for p in (1, 4):      # the U points between the two CENTER columns: the ones whose candidate quads include the only quad
            assert sorted(idx[p].tolist()) == [0, 1, 2, 3], (c["scale_km"], idx[p])
            got = np.empty(4)
            got[idx[p]] = w[p]
            err = np.abs(got - np.array(c["w"][p])).max()
            worst[c["scale_km"]] = max(worst.get(c["scale_km"], 0.0), err)
    assert max(worst.values()) < 1e-12, worst
