#!/usr/bin/env python3
"""Generator of tests/golden/weights_hp.json -- high-precision known answers for the two weight formulas of the hot path.

The reference delegates them to ESMF (absent here), so these goldens do not pin ESMF; they pin the ARITHMETIC of the oracle
(and, through the parity tests, of the kernels) to the mathematical definitions SURVEY App. A2 / A5 state, evaluated by a
DIFFERENT route at 50 digits (mpmath):

* bilinear on a dual triangle (A2): the weights of target point P in triangle (A, B, C) of unit vectors are the solution of
  A + u (B - A) + v (C - A) = t P, w = (1 - u - v, u, v) -- here by solving the 3 x 3 linear system for (u, v, t) (the code
  under test uses the determinant closed form in difference form); and `w_normal`, the barycentric coordinates of the foot
  of P on the triangle's plane along the plane's normal -- the other reading of straight cell edges on a sphere
  ("bilinear_linetype" 1), by a second 3 x 3 solve;
* first-order conservative (A5): w = Area(src ^ dst) / Area(dst) with great-circle sides -- here the intersection polygon
  from Sutherland-Hodgman clipping carried out at 50 digits, and areas from GIRARD's theorem (sum of the interior angles
  minus (n - 2) pi) instead of the code's triangle fan with the Van Oosterom-Strackee formula.

* grid -> grid bilinear on the quad of four CENTER points (A4, the U / V destaggering): the bilinear map
  X(s, t) = lam P solved by mpmath.findroot at 40 digits (the code under test runs its own Newton iteration in float64).

Cases span the scales of the BASELINE configurations (3-km to 500-km cells) because the difference forms exist for the small
ones.  Output: inputs as float64 (exactly what the tests feed the oracle) and expected weights as float64 roundings of the
50-digit results.

    python tests/golden/make_weight_goldens.py        # rewrites tests/golden/weights_hp.json (deterministic)
"""
import json
import os
import random

import mpmath as mp

mp.mp.dps = 50


def unit(v):
    n = mp.sqrt(sum(x * x for x in v))
    return [x / n for x in v]


def from_lonlat(lon, lat):
    lon, lat = mp.mpf(lon) * mp.pi / 180, mp.mpf(lat) * mp.pi / 180
    return [mp.cos(lat) * mp.cos(lon), mp.cos(lat) * mp.sin(lon), mp.sin(lat)]


def as_f64(v):
    return [float(x) for x in v]


def lift(v):
    """float64 triple -> mpf triple EXACTLY as stored (the tests pass these float64 values on)."""
    return [mp.mpf(x) for x in v]


def cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def dot(a, b):
    return sum(x * y for x, y in zip(a, b))


def sub(a, b):
    return [x - y for x, y in zip(a, b)]


# ---- bilinear: solve [B-A, C-A, -P] (u, v, t)^T = -A -----------------------------------------------------------------
def tri_weights(A, B, C, P):
    M = mp.matrix([[B[0] - A[0], C[0] - A[0], -P[0]], [B[1] - A[1], C[1] - A[1], -P[1]], [B[2] - A[2], C[2] - A[2], -P[2]]])
    rhs = mp.matrix([-A[0], -A[1], -A[2]])
    u, v, t = mp.lu_solve(M, rhs)
    return [1 - u - v, u, v], t


# ---- the same triangle with the point dropped along the plane's NORMAL (ESMF_LINETYPE_CART read literally; the code's
# "bilinear_linetype" 1): solve [B-A, C-A, n] (u, v, s)^T = P - A with n = (B - A) x (C - A)
def tri_weights_normal(A, B, C, P):
    n = cross(sub(B, A), sub(C, A))
    M = mp.matrix([[B[0] - A[0], C[0] - A[0], n[0]], [B[1] - A[1], C[1] - A[1], n[1]], [B[2] - A[2], C[2] - A[2], n[2]]])
    rhs = mp.matrix([P[0] - A[0], P[1] - A[1], P[2] - A[2]])
    u, v, _ = mp.lu_solve(M, rhs)
    return [1 - u - v, u, v]


# ---- conservative: great-circle clipping + Girard ----------------------------------------------------------------------
def clip(poly, a, b):
    """keep the part of `poly` on the left of the great circle a -> b (normal a x b)"""
    n = cross(a, b)
    out = []
    m = len(poly)
    for k in range(m):
        X1, X2 = poly[k], poly[(k + 1) % m]
        d1, d2 = dot(n, X1), dot(n, X2)
        if d1 >= 0:
            out.append(X1)
        if (d1 >= 0) != (d2 >= 0):
            X = [x1 * d2 - x2 * d1 for x1, x2 in zip(X1, X2)]
            if d2 - d1 < 0:
                X = [-x for x in X]
            out.append(unit(X))
    return out


def girard_area(poly):
    n = len(poly)
    if n < 3:
        return mp.mpf(0)
    total = mp.mpf(0)
    for k in range(n):
        prv, cur, nxt = poly[k - 1], poly[k], poly[(k + 1) % n]
        # tangent directions at `cur` towards its neighbours
        t1 = sub(prv, [c * dot(prv, cur) for c in cur])
        t2 = sub(nxt, [c * dot(nxt, cur) for c in cur])
        n1, n2 = mp.sqrt(dot(t1, t1)), mp.sqrt(dot(t2, t2))
        if n1 == 0 or n2 == 0:
            continue
        c = dot(t1, t2) / (n1 * n2)
        c = max(min(c, 1), -1)
        total += mp.acos(c)
    return total - (n - 2) * mp.pi


def dedupe(poly):
    out = []
    for p in poly:
        if not out or mp.sqrt(dot(sub(p, out[-1]), sub(p, out[-1]))) > mp.mpf(10) ** -40:
            out.append(p)
    if len(out) > 1 and mp.sqrt(dot(sub(out[0], out[-1]), sub(out[0], out[-1]))) <= mp.mpf(10) ** -40:
        out.pop()
    return out


def main():
    rng = random.Random(4711)
    bil = []
    for scale_km in (3.0, 30.0, 120.0, 500.0):
        for _ in range(40):
            lon0, lat0 = rng.uniform(-180, 180), rng.uniform(-85, 85)
            d = scale_km / 111.0                                  # degrees
            tri = []
            for k in range(3):
                ang = 2 * 3.141592653589793 * (k / 3.0 + rng.uniform(-0.08, 0.08))
                r = d * rng.uniform(0.6, 1.0)
                tri.append(as_f64(unit(from_lonlat(lon0 + r * float(mp.cos(ang)) / max(0.1, float(mp.cos(mp.mpf(lat0) * mp.pi / 180))),
                                                   lat0 + r * float(mp.sin(ang))))))
            # a point inside: random convex combination, pushed onto the sphere, stored as float64
            a, b = rng.uniform(0.05, 0.9), rng.uniform(0.05, 0.9)
            if a + b > 0.95:
                a, b = a * 0.5, b * 0.5
            A, B, C = (lift(v) for v in tri)
            P = as_f64(unit([A[i] * (1 - a - b) + B[i] * a + C[i] * b for i in range(3)]))
            w, t = tri_weights(A, B, C, lift(P))
            assert t > 0 and all(x > 0 for x in w)
            wn = tri_weights_normal(A, B, C, lift(P))
            bil.append(dict(scale_km=scale_km, tri=tri, p=P, w=as_f64(w), w_normal=as_f64(wn)))
    cons = []
    for scale_km, ratio in ((3.0, 1.0), (30.0, 0.1), (30.0, 1.3), (120.0, 0.4), (500.0, 0.05)):
        for _ in range(6):
            lon0, lat0 = rng.uniform(-180, 180), rng.uniform(-80, 80)
            d = scale_km / 111.0
            coslat = max(0.15, float(mp.cos(mp.mpf(lat0) * mp.pi / 180)))
            nv = rng.choice((5, 6, 6, 7))
            src = []
            for k in range(nv):                                   # a convex, counter-clockwise polygon around (lon0, lat0)
                ang = 2 * 3.141592653589793 * (k + rng.uniform(-0.15, 0.15)) / nv
                r = 0.5 * d * rng.uniform(0.85, 1.0)
                src.append(as_f64(unit(from_lonlat(lon0 + r * float(mp.cos(ang)) / coslat, lat0 + r * float(mp.sin(ang))))))
            # a 4 x 4 patch of destination cells of size ratio * source size, slightly rotated, roughly centred on the polygon
            q = d * ratio
            rot = rng.uniform(-0.3, 0.3)
            cr, sr = float(mp.cos(rot)), float(mp.sin(rot))
            ox, oy = rng.uniform(-0.5, 0.5) * q, rng.uniform(-0.5, 0.5) * q
            corners = []
            for jj in range(5):
                for ii in range(5):
                    x, y = (ii - 2) * q + ox, (jj - 2) * q + oy
                    corners.append(as_f64(unit(from_lonlat(lon0 + (cr * x - sr * y) / coslat, lat0 + (sr * x + cr * y)))))
            S = [lift(v) for v in src]
            rows = []
            for jj in range(4):
                for ii in range(4):
                    quad = [lift(corners[jj * 5 + ii]), lift(corners[jj * 5 + ii + 1]), lift(corners[(jj + 1) * 5 + ii + 1]),
                            lift(corners[(jj + 1) * 5 + ii])]
                    poly = list(S)
                    for e in range(4):
                        poly = clip(poly, quad[e], quad[(e + 1) % 4])
                        if len(poly) < 3:
                            break
                    poly = dedupe(poly)
                    ai = girard_area(poly) if len(poly) >= 3 else mp.mpf(0)
                    aq = girard_area(quad)
                    rows.append(float(ai / aq) if ai > 0 else 0.0)
            cons.append(dict(scale_km=scale_km, ratio=ratio, src=src, corners=corners, w=rows, src_area=float(girard_area(S))))
    # ---- grid -> grid bilinear on a quad of four CENTER points (A4): X(s, t) = A + s (B - A) + t (D - A) + s t (A - B + C - D) = lam P
    quads = []
    for scale_km in (3.0, 12.0, 60.0, 300.0):
        for _ in range(10):
            lon0, lat0 = rng.uniform(-180, 180), rng.uniform(-80, 80)
            d = scale_km / 111.0
            coslat = max(0.15, float(mp.cos(mp.mpf(lat0) * mp.pi / 180)))
            rot = rng.uniform(-0.5, 0.5)
            cr, sr = float(mp.cos(rot)), float(mp.sin(rot))
            cen = []
            for (x, y) in ((0, 0), (1, 0), (0, 1), (1, 1)):      # flat index j * 2 + i of a 2 x 2 CENTER grid
                xx, yy = (x + rng.uniform(-0.05, 0.05)) * d, (y + rng.uniform(-0.05, 0.05)) * d
                cen.append(as_f64(unit(from_lonlat(lon0 + (cr * xx - sr * yy) / coslat, lat0 + (sr * xx + cr * yy)))))
            A, B, D, Cq = (lift(v) for v in cen)
            pts, ws = [], []
            for _k in range(6):                                    # EDGE1 stagger of a 2 x 2 grid has 3 x 2 points
                s0, t0 = rng.uniform(0.05, 0.95), rng.uniform(0.05, 0.95)
                X = [A[i] + s0 * (B[i] - A[i]) + t0 * (D[i] - A[i]) + s0 * t0 * (A[i] - B[i] + Cq[i] - D[i]) for i in range(3)]
                P = as_f64(unit(X))
                Pm = lift(P)
                f = lambda s_, t_, l_: [A[i] + s_ * (B[i] - A[i]) + t_ * (D[i] - A[i]) + s_ * t_ * (A[i] - B[i] + Cq[i] - D[i]) - l_ * Pm[i]
                                        for i in range(3)]
                sol = mp.findroot(f, (mp.mpf(s0), mp.mpf(t0), mp.mpf(1)), tol=mp.mpf(10) ** -40, maxsteps=50)
                s1, t1 = sol[0], sol[1]
                pts.append(P)
                ws.append(as_f64([(1 - s1) * (1 - t1), s1 * (1 - t1), (1 - s1) * t1, s1 * t1]))   # weights of flat centres 0..3
            quads.append(dict(scale_km=scale_km, centres=cen, pts=pts, w=ws))
    out = dict(note="generated by tests/golden/make_weight_goldens.py (mpmath, 50 digits); see its docstring", bilinear=bil, conserve=cons,
               quad=quads)
    import sys
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "weights_hp.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print("wrote %s: %d triangles, %d polygon / 4x4-grid cases, %d quads x 6 points" % (path, len(bil), len(cons), len(quads)))


if __name__ == "__main__":
    main()
