#!/usr/bin/env python3
"""Generator of tests/golden/store_hp.json -- WHOLE RegridStore results by brute force at 50 digits (mpmath).

weights_hp.json pins the two weight FORMULAS on single triangles / polygons.  This fixture pins what surrounds them in a Store: which
dual triangle holds a target point (search, rim, unmapped points), which cell is nearest, which (source cell, destination cell) pairs
overlap and by how much, how a matrix row is normalised -- on three small meshes and grids, by a route that shares nothing with the code
under test except the definitions SURVEY App. A states:

* element-located bilinear (A2): the dual triangle of every mesh vertex touched by exactly three cells = their centres; the triangle that
  holds target point P is the one for which  A + u (B - A) + v (C - A) = t P  has u, v, 1 - u - v >= 0 and t > 0 (3 x 3 solve at 50 digits
  for EVERY candidate triangle, no search structure); weights (1 - u - v, u, v); a point in no triangle is unmapped (IGNORE);
* nearest source to destination (A3): argmin over ALL cell centres of the chord distance;
* Grid -> Grid destaggering (A4): see grid_to_grid below;
* first-order conservative (A5): for every destination cell (quad of four CORNER points) and every source cell (polygon of its vertices)
  the great-circle intersection polygon by Sutherland-Hodgman at 50 digits, areas by Girard's theorem, w = area(src ^ dst) / area(dst).

Coordinates start from the float64 latitudes / longitudes the tests hand to the library (radians for the mesh, degrees for the grid),
converted to unit vectors at 50 digits; the float64 conversion inside the library / oracle costs ~1e-16 x (radius / cell size) in a weight.
The only float64 step here is a generous distance pre-filter (candidates beyond the sum of both shapes' bounding radii cannot meet).
Points closer than 1e-9 (in barycentric units / in relative distance) to a tie are marked `tie` and skipped by the tests.

    python tests/golden/make_store_goldens.py        # rewrites tests/golden/store_hp.json (deterministic; ~10 s)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import mpmath as mp  # noqa: E402
from make_weight_goldens import clip, cross, dedupe, dot, girard_area, sub, tri_weights  # noqa: E402

mp.mp.dps = 50


def xyz_from_rad(lon, lat):
    lon, lat = mp.mpf(float(lon)), mp.mpf(float(lat))
    return [mp.cos(lat) * mp.cos(lon), mp.cos(lat) * mp.sin(lon), mp.sin(lat)]


def xyz_from_deg(lon, lat):
    return xyz_from_rad_mp(mp.mpf(float(lon)) * mp.pi / 180, mp.mpf(float(lat)) * mp.pi / 180)


def xyz_from_rad_mp(lon, lat):
    return [mp.cos(lat) * mp.cos(lon), mp.cos(lat) * mp.sin(lon), mp.sin(lat)]


def f3(v):
    return np.array([float(x) for x in v])


def ccw(poly):
    """orientation seen from outside: sign of sum of (P_k x P_k+1) . centroid"""
    c = [sum(p[i] for p in poly) for i in range(3)]
    s = mp.mpf(0)
    for k in range(len(poly)):
        s += dot(cross(poly[k], poly[(k + 1) % len(poly)]), c)
    return poly if s > 0 else poly[::-1]


def store(mesh, lon_c, lat_c, lon_k, lat_k):
    """mesh: MpasMesh (radians); lon_c / lat_c [ny][nx] centres, lon_k / lat_k [ny+1][nx+1] corners (degrees)."""
    nC, nV = mesh.nCells, mesh.nVertices
    C = [xyz_from_rad(mesh.lonCell[i], mesh.latCell[i]) for i in range(nC)]
    V = [xyz_from_rad(mesh.lonVertex[i], mesh.latVertex[i]) for i in range(nV)]
    Cf, Vf = np.array([f3(c) for c in C]), np.array([f3(v) for v in V])
    voc = mesh.verticesOnCell
    cov = [[] for _ in range(nV)]                        # cells on vertex
    for c in range(nC):
        for v in voc[c]:
            if v > 0:
                cov[v - 1].append(c)
    tris = [tuple(cs) for cs in cov if len(cs) == 3]
    ny, nx = lat_c.shape
    P = [xyz_from_deg(lon_c.flat[p], lat_c.flat[p]) for p in range(nx * ny)]
    Pf = np.array([f3(p) for p in P])
    # ---- bilinear -------------------------------------------------------------------------------------------------------------
    tri_r = np.array([max(np.linalg.norm(Cf[a] - Cf[b]), np.linalg.norm(Cf[b] - Cf[c]), np.linalg.norm(Cf[a] - Cf[c])) for a, b, c in tris])
    tri_c = np.array([Cf[a] for a, _, _ in tris])
    bil = []
    for p in range(nx * ny):
        near = np.nonzero(np.linalg.norm(tri_c - Pf[p], axis=1) <= 1.5 * tri_r + 1e-9)[0]    # P inside => within the longest side (+ the sagitta) of vertex A
        hits = []
        for k in near:
            a, b, c = tris[k]
            w, t = tri_weights(C[a], C[b], C[c], P[p])
            if t > 0 and min(w) >= -mp.mpf(10) ** -9:
                hits.append((min(w), (a, b, c), w))
        if not hits:
            bil.append(None)
            continue
        inside = [h for h in hits if h[0] > mp.mpf(10) ** -9]
        if len(hits) > 1 or not inside:                  # on (or within 1e-9 of) an edge or a vertex: the owner is a tie rule's business
            bil.append({"tie": True})
            continue
        _, ids, w = inside[0]
        order = sorted(range(3), key=lambda k: ids[k])
        bil.append({"col": [int(ids[k]) for k in order], "w": [float(w[k]) for k in order]})
    # ---- nearest --------------------------------------------------------------------------------------------------------------
    nst = []
    for p in range(nx * ny):
        d = np.linalg.norm(Cf - Pf[p], axis=1)
        cand = np.nonzero(d <= d.min() * (1 + 1e-6) + 1e-12)[0]
        dd = sorted((mp.sqrt(dot(sub(C[c], P[p]), sub(C[c], P[p]))), int(c)) for c in cand)
        tie = len(dd) > 1 and (dd[1][0] - dd[0][0]) <= mp.mpf(10) ** -9 * dd[0][0]
        nst.append({"tie": True} if tie else {"col": dd[0][1]})
    # ---- conservative ---------------------------------------------------------------------------------------------------------
    K = [[xyz_from_deg(lon_k[j, i], lat_k[j, i]) for i in range(nx + 1)] for j in range(ny + 1)]
    polys, prad = [], []
    for c in range(nC):
        poly = ccw([V[v - 1] for v in voc[c] if v > 0])
        polys.append(poly)
        prad.append(max(np.linalg.norm(f3(q) - Cf[c]) for q in poly))
    cons = []
    for j in range(ny):
        for i in range(nx):
            quad = ccw([K[j][i], K[j][i + 1], K[j + 1][i + 1], K[j + 1][i]])
            qf = np.array([f3(q) for q in quad])
            qc = qf.mean(axis=0)
            qr = np.linalg.norm(qf - qc, axis=1).max()
            aq = girard_area(quad)
            for c in np.nonzero(np.linalg.norm(Cf - qc, axis=1) <= (np.array(prad) + qr) * 1.05 + 1e-9)[0]:
                poly = list(polys[c])
                for e in range(4):
                    poly = clip(poly, quad[e], quad[(e + 1) % 4])
                    if len(poly) < 3:
                        break
                poly = dedupe(poly)
                if len(poly) < 3:
                    continue
                a = girard_area(poly)
                if a > 0:
                    cons.append([j * nx + i, int(c), float(a / aq)])
    return {"bilinear": bil, "nearest": nst, "conserve": cons}


def grid_to_grid(lon_c, lat_c, lon_e, lat_e):
    """Destaggering (A4): every EDGE point in the bilinear quad of four CENTER points that holds it -- X(s, t) = A + s (B - A) + t (D - A) +
    s t (A - B + C - D) = lam P solved by findroot at 40 digits for every quad near the point; a point held by no quad is unmapped.  A C-grid's
    edge points lie (to the grid's curvature) ON the line between two centres, i.e. on the border of two quads: both give the same sparse
    weight vector, which is what is recorded ({centre: weight}, weights below 1e-13 dropped)."""
    ny, nx = lat_c.shape
    Cm = [[xyz_from_deg(lon_c[j, i], lat_c[j, i]) for i in range(nx)] for j in range(ny)]
    Cf = np.array([[f3(Cm[j][i]) for i in range(nx)] for j in range(ny)])
    qc = 0.25 * (Cf[:-1, :-1] + Cf[:-1, 1:] + Cf[1:, :-1] + Cf[1:, 1:])                      # quad centres (float64: the pre-filter only)
    qr = np.linalg.norm(Cf[:-1, :-1] - Cf[1:, 1:], axis=2)
    out = []
    for p in range(lat_e.size):
        P = xyz_from_deg(lon_e.flat[p], lat_e.flat[p])
        Pf = f3(P)
        cand = np.argwhere(np.linalg.norm(qc - Pf, axis=2) <= 1.2 * qr)
        vec, tie, near, strict = None, False, False, False
        for j, i in cand:
            A, B, D, Cq = Cm[j][i], Cm[j][i + 1], Cm[j + 1][i], Cm[j + 1][i + 1]
            f = lambda s_, t_, l_: [A[k] + s_ * (B[k] - A[k]) + t_ * (D[k] - A[k]) + s_ * t_ * (A[k] - B[k] + Cq[k] - D[k]) - l_ * P[k] for k in range(3)]   # noqa: E731
            try:
                sol = mp.findroot(f, (mp.mpf("0.5"), mp.mpf("0.5"), mp.mpf(1)), tol=mp.mpf(10) ** -40, maxsteps=60)
            except (ValueError, ZeroDivisionError):
                continue
            s1, t1, lam = sol[0], sol[1], sol[2]
            eps = mp.mpf(10) ** -9
            if lam <= 0:
                continue
            out_by = max(-s1, s1 - 1, -t1, t1 - 1)                   # > 0: that far outside this quad (in its own s, t)
            if out_by > eps:
                if out_by < mp.mpf(10) ** -6:
                    near = True                                       # a hair outside: on the hull, or over the border into a neighbour
                continue
            w = {int(j * nx + i): (1 - s1) * (1 - t1), int(j * nx + i + 1): s1 * (1 - t1), int((j + 1) * nx + i): (1 - s1) * t1,
                 int((j + 1) * nx + i + 1): s1 * t1}
            w = {k: float(v) for k, v in w.items() if abs(v) > mp.mpf(10) ** -13}
            strict = strict or out_by < -eps
            if vec is None:
                vec = w
            else:                                                     # a second quad holds the point (within 1e-9 of their shared border): the two
                keys = set(vec) | set(w)                              # vectors differ by the distance from the border at most -- marked, either is right
                assert all(abs(vec.get(k, 0.0) - w.get(k, 0.0)) < 1e-8 for k in keys), (p, vec, w)
                tie = True
        if vec is None:                                               # in no quad; `hull`: within 1e-6 of one (on the hull of the CENTER points to rounding:
            out.append({"hull": True} if near else None)              # whether such a point is mapped is a tolerance's business, not the geometry's)
        elif not strict and not tie:                                  # held by ONE quad and only within 1e-9 of its border: the hull again
            out.append({"hull": True})
        else:
            out.append(dict({"col": sorted(vec), "w": [vec[k] for k in sorted(vec)]}, **({"tie": True} if tie else {})))
    return out


def case(name, mesh, g):
    print("case %s: %d cells, %d vertices, grid %d x %d" % (name, mesh.nCells, mesh.nVertices, g.nx, g.ny), flush=True)
    res = store(mesh, g.lon, g.lat, g.lon_c, g.lat_c)
    res["edge1"] = grid_to_grid(g.lon, g.lat, g.lon_u, g.lat_u)
    res["edge2"] = grid_to_grid(g.lon, g.lat, g.lon_v, g.lat_v)
    cnt = lambda k: (sum(1 for e in res[k] if e and "col" in e), sum(e is None for e in res[k]), sum(1 for e in res[k] if e and "hull" in e))   # noqa: E731
    print("  grid -> grid: EDGE1 %d mapped / %d unmapped / %d on the hull, EDGE2 %d / %d / %d" % (cnt("edge1") + cnt("edge2")), flush=True)
    nb = sum(1 for b in res["bilinear"] if b and "col" in b)
    print("  bilinear: %d mapped, %d unmapped, %d ties; nearest ties %d; conservative entries %d" % (
        nb, sum(1 for b in res["bilinear"] if b is None), sum(1 for b in res["bilinear"] if b and "tie" in b),
        sum(1 for b in res["nearest"] if "tie" in b), len(res["conserve"])), flush=True)
    return {"name": name, "nx": int(g.nx), "ny": int(g.ny),
            "latCell": mesh.latCell.tolist(), "lonCell": mesh.lonCell.tolist(), "latVertex": mesh.latVertex.tolist(), "lonVertex": mesh.lonVertex.tolist(),
            "verticesOnCell": mesh.verticesOnCell.tolist(),
            "lon": g.lon.ravel().tolist(), "lat": g.lat.ravel().tolist(), "lon_c": g.lon_c.ravel().tolist(), "lat_c": g.lat_c.ravel().tolist(),
            "lon_u": g.lon_u.ravel().tolist(), "lat_u": g.lat_u.ravel().tolist(), "lon_v": g.lon_v.ravel().tolist(), "lat_v": g.lat_v.ravel().tolist(),
            "expect": res}


def main():
    from mpassit_amd import synth, target_grid as tg
    cases = []
    # 1. a regional hexagon mesh under a Lambert grid that is LARGER than the mesh (rim, unmapped points, partly covered cells)
    g = tg.define_target_grid_params("lambert", 18, 14, dx=120000.0, dy=120000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5)
    m = synth.regional_mesh_for_lambert(g.proj, 15, 11, 320, margin=0.0)
    cases.append(case("regional_lambert_120km", m, g))
    # 2. a global Voronoi mesh (pentagons .. heptagons) under a regional lat-lon grid that crosses the date line
    m2 = synth.global_voronoi_mesh(260)
    g2 = tg.define_target_grid_params("lat-lon", 13, 9, dx=7.0, dy=7.0, ref_lat=5.0, ref_lon=178.0)
    cases.append(case("global_voronoi_dateline_latlon_7deg", m2, g2))
    # 3. the same mesh under a polar stereographic grid centred on the North Pole (the pole inside a cell of either kind)
    g3 = tg.define_target_grid_params("polar", 10, 10, dx=500000.0, dy=500000.0, ref_lat=90.0, ref_lon=0.0, truelat1=60.0, stand_lon=-30.0)
    cases.append(case("global_voronoi_north_pole_ps_500km", m2, g3))
    # 4. case 1's shapes at the 3-km scale of the headline configuration (where the difference forms of the weight formulas matter)
    g4 = tg.define_target_grid_params("lambert", 18, 14, dx=3000.0, dy=3000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5)
    m4 = synth.regional_mesh_for_lambert(g4.proj, 15, 11, 320, margin=0.0)
    cases.append(case("regional_lambert_3km", m4, g4))
    out = {"note": "generated by tests/golden/make_store_goldens.py (mpmath, 50 digits, brute force); see its docstring", "cases": cases}
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "store_hp.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print("wrote %s (%d bytes)" % (path, os.path.getsize(path)))


if __name__ == "__main__":
    main()
