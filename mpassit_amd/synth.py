"""Deterministic synthetic MPAS-like meshes and fields (seed 20240807).

No MPAS files or NetCDF exist in the build environment, so tests and bench.py use:
  * global quasi-uniform Voronoi meshes from `scipy.spatial.SphericalVoronoi` (real unstructured
    connectivity: pentagons/hexagons/heptagons) for the small parity cases, and
  * regional perturbed-hexagonal meshes laid out in a Lambert plane (O(N) to generate, exact
    Voronoi vertices = spherical circumcentres) for the 655 362-cell / 3 M-cell bench workloads.
Arrays follow the MPAS file conventions the reference reads (model_grid.F90:354-417): lat/lon in
radians, lon in [0, 2pi), `verticesOnCell` [nCells][maxEdges] int32, 1-based, 0-padded.
"""
from dataclasses import dataclass

import numpy as np

SEED = 20240807


@dataclass
class MpasMesh:
    latCell: np.ndarray
    lonCell: np.ndarray
    latVertex: np.ndarray
    lonVertex: np.ndarray
    verticesOnCell: np.ndarray  # [nCells][maxEdges] int32, 1-based, 0 = pad

    @property
    def nCells(self):
        return self.latCell.size

    @property
    def nVertices(self):
        return self.latVertex.size

    @property
    def maxEdges(self):
        return self.verticesOnCell.shape[1]


def _xyz_to_latlon_rad(xyz):
    lat = np.arcsin(np.clip(xyz[:, 2], -1.0, 1.0))
    lon = np.arctan2(xyz[:, 1], xyz[:, 0])
    lon = np.where(lon < 0.0, lon + 2.0 * np.pi, lon)
    return lat, lon


def latlon_rad_to_xyz(lat, lon):
    cl = np.cos(lat)
    return np.stack([cl * np.cos(lon), cl * np.sin(lon), np.sin(lat)], axis=-1)


def global_voronoi_mesh(n_cells, seed=SEED, jitter=0.25, max_edges=None):
    """Quasi-uniform global mesh: jittered spherical-Fibonacci generators + SphericalVoronoi."""
    from scipy.spatial import SphericalVoronoi
    rng = np.random.default_rng(seed)
    k = np.arange(n_cells) + 0.5
    phi = np.arccos(1.0 - 2.0 * k / n_cells)
    theta = np.pi * (1.0 + 5.0 ** 0.5) * k
    pts = np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=1)
    h = np.sqrt(4.0 * np.pi / n_cells)
    pts = pts + jitter * h * rng.uniform(-0.5, 0.5, pts.shape)
    pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    sv = SphericalVoronoi(pts, radius=1.0, center=np.zeros(3))
    sv.sort_vertices_of_regions()
    me = max(len(r) for r in sv.regions)
    if max_edges is None:
        max_edges = me
    assert max_edges >= me
    voc = np.zeros((n_cells, max_edges), np.int32)
    for c, r in enumerate(sv.regions):
        r = np.asarray(r, np.int32)
        # CCW seen from outside (MPAS convention)
        a, b, cc = sv.vertices[r[0]], sv.vertices[r[1]], sv.vertices[r[2]]
        if np.dot(pts[c], np.cross(b - a, cc - a)) < 0:
            r = r[::-1]
        voc[c, :len(r)] = r + 1
    latc, lonc = _xyz_to_latlon_rad(pts)
    latv, lonv = _xyz_to_latlon_rad(sv.vertices)
    return MpasMesh(latc, lonc, latv, lonv, voc)


def _morton_order(xyz, bits=10):
    """Permutation that sorts unit vectors along a 3-D Morton curve (locality-preserving numbering)."""
    q = np.clip(((xyz + 1.0) * 0.5 * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(xyz.shape[0], np.int64)
    for b in range(bits):
        for d in range(3):
            code |= ((q[:, d] >> b) & 1) << (3 * b + d)
    return np.argsort(code, kind="stable")


def icosahedral_mesh(level, order="morton"):
    """Global quasi-uniform mesh by `level` bisections of the icosahedron, the construction behind the MPAS x1 meshes:
    10*4^level + 2 cells (level 6: 40 962, 8: 655 362, 9: 2 621 442), 12 pentagons, hexagons elsewhere; the Voronoi
    vertices are the circumcentres of the bisection triangles.  O(N log N) numpy, no scipy.
    order="morton": cells and vertices renumbered along a space-filling curve (production meshes are reordered for
    locality too); order="native": bisection order (each level's midpoints appended), which is close to random."""
    verts, faces = _icosahedron()
    for _ in range(level):
        nv, nf = verts.shape[0], faces.shape[0]
        e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
        key = np.minimum(e[:, 0], e[:, 1]) * nv + np.maximum(e[:, 0], e[:, 1])
        uk, inv = np.unique(key, return_inverse=True)
        mid = verts[uk // nv] + verts[uk % nv]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        m = inv.reshape(-1) + nv
        m01, m12, m20 = m[:nf], m[nf:2 * nf], m[2 * nf:]
        v0, v1, v2 = faces[:, 0], faces[:, 1], faces[:, 2]
        faces = np.concatenate([np.stack([v0, m01, m20], 1), np.stack([v1, m12, m01], 1), np.stack([v2, m20, m12], 1),
                                np.stack([m01, m12, m20], 1)])
        verts = np.concatenate([verts, mid])
    return _mesh_from_sphere_triangulation(verts, faces, order)


def _icosahedron():
    t = (1.0 + 5.0 ** 0.5) / 2.0
    verts = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                      [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    verts /= np.linalg.norm(verts, axis=1, keepdims=True)
    faces = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                      [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                      [8, 6, 7], [9, 8, 1]], np.int64)
    return verts, faces


def geodesic_mesh(freq, order="morton"):
    """Global quasi-uniform mesh of ANY size class: every face of the icosahedron divided into freq^2 triangles (a class-I geodesic
    grid) -> 10 * freq^2 + 2 cells, 12 pentagons, hexagons elsewhere -- freq = 2^level reproduces icosahedral_mesh's counts (not its
    points: bisection re-projects at every level).  freq = 548: 3 003 042 cells, the "3 M-cell" global mesh of BASELINE configs[4]
    (the bisection family jumps from 2.6 M to 10.5 M).  O(N log N) numpy."""
    if freq < 1:
        raise ValueError("freq >= 1")
    iv, ifc = _icosahedron()
    n = int(freq)
    # lattice points of one face: (i, j) with i + j <= n  ->  (i * B + j * C + (n - i - j) * A) / n
    ii, jj = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="ij")
    keep = ii + jj <= n
    ii, jj = ii[keep], jj[keep]
    kk = n - ii - jj
    local = -np.ones((n + 1, n + 1), np.int64)
    local[ii, jj] = np.arange(ii.size)
    # triangles of the face lattice: upward (i,j),(i+1,j),(i,j+1) and downward (i+1,j),(i+1,j+1),(i,j+1)
    ui, uj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    up = ui + uj <= n - 1
    dn = ui + uj <= n - 2
    tri_l = np.concatenate([np.stack([local[ui[up], uj[up]], local[ui[up] + 1, uj[up]], local[ui[up], uj[up] + 1]], 1),
                            np.stack([local[ui[dn] + 1, uj[dn]], local[ui[dn] + 1, uj[dn] + 1], local[ui[dn], uj[dn] + 1]], 1)])
    pts, tris = [], []
    npf = ii.size
    for f, (a, b, c) in enumerate(ifc):
        p = (kk[:, None] * iv[a] + ii[:, None] * iv[b] + jj[:, None] * iv[c]) / float(n)
        pts.append(p / np.linalg.norm(p, axis=1, keepdims=True))
        tris.append(tri_l + f * npf)
    pts, tris = np.concatenate(pts), np.concatenate(tris)
    # points on shared edges / corners come once per face: merge them (they agree to rounding; the lattice spacing is ~1 / freq)
    key = np.round(pts * (1 << 26)).astype(np.int64)
    _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
    verts = pts[first]
    faces = inv.reshape(-1)[tris]
    if verts.shape[0] != 10 * n * n + 2 or faces.shape[0] != 20 * n * n:
        raise RuntimeError("geodesic_mesh: %d points / %d triangles, expected %d / %d" % (verts.shape[0], faces.shape[0], 10 * n * n + 2, 20 * n * n))
    return _mesh_from_sphere_triangulation(verts, faces, order)


def _mesh_from_sphere_triangulation(verts, faces, order):
    """Delaunay triangulation of the sphere (points = cell centres, triangles) -> MpasMesh: Voronoi vertices = spherical circumcentres,
    verticesOnCell counter-clockwise; order = 'morton' (cells and vertices along a space-filling curve) | 'native'."""
    faces = faces.copy()
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    n = np.cross(b - a, c - a)
    flip = np.einsum("ij,ij->i", n, a + b + c) < 0.0          # counter-clockwise seen from outside
    faces[flip] = faces[flip][:, [0, 2, 1]]
    n[flip] *= -1.0
    cc = n / np.linalg.norm(n, axis=1, keepdims=True)          # spherical circumcentres = Voronoi vertices
    del a, b, c, n
    if order == "morton":
        pc = _morton_order(verts)
        rank = np.empty_like(pc)
        rank[pc] = np.arange(pc.size)
        verts, faces = verts[pc], rank[faces]
        pt = _morton_order(cc)
        cc, faces = cc[pt], faces[pt]
    elif order != "native":
        raise ValueError("order must be 'morton' or 'native'")
    # verticesOnCell: the triangles around each cell sorted by azimuth in the cell's tangent plane (CCW)
    nc, nt = verts.shape[0], faces.shape[0]
    cell = faces.reshape(-1)
    tri = np.repeat(np.arange(nt, dtype=np.int64), 3)
    p = verts[cell]
    ref = np.where((np.abs(p[:, 2]) < 0.9)[:, None], np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.0, 0.0]))
    e1 = ref - np.einsum("ij,ij->i", ref, p)[:, None] * p
    e2 = np.cross(p, e1)
    d = cc[tri] - p
    ang = np.arctan2(np.einsum("ij,ij->i", d, e2), np.einsum("ij,ij->i", d, e1))
    del p, ref, e1, e2, d
    o = np.lexsort((ang, cell))
    cell, tri = cell[o], tri[o]
    start = np.concatenate([[0], np.cumsum(np.bincount(cell, minlength=nc))])
    voc = np.zeros((nc, 6), np.int32)
    voc[cell, np.arange(cell.size) - start[cell]] = tri + 1
    latc, lonc = _xyz_to_latlon_rad(verts)
    latv, lonv = _xyz_to_latlon_rad(cc)
    return MpasMesh(latc, lonc, latv, lonv, voc)


def variable_resolution_mesh(n_cells, lat0_deg=38.5, lon0_deg=-97.5, ratio=8.0, radius_deg=25.0, seed=SEED):
    """Global variable-resolution Voronoi mesh (like the MPAS 60-3 km / 15-3 km meshes): generator density is
    `ratio`^2 times higher inside a cap of `radius_deg` around (lat0, lon0) than far away, with a smooth transition.
    Cells are random (Poisson-like) points thinned by the density function -- irregular pentagons..octagons, strongly
    varying triangle sizes: a stress test for the search structures."""
    from scipy.spatial import SphericalVoronoi
    rng = np.random.default_rng(seed + 3)
    c = latlon_rad_to_xyz(np.deg2rad(lat0_deg), np.deg2rad(lon0_deg))
    pts = np.empty((0, 3))
    while pts.shape[0] < n_cells:
        cand = rng.standard_normal((4 * n_cells, 3))
        cand /= np.linalg.norm(cand, axis=1, keepdims=True)
        ang = np.degrees(np.arccos(np.clip(cand @ c, -1, 1)))
        # spacing grows from 1 (inside) to `ratio` (outside) over one more radius; density ~ 1/spacing^2
        spacing = 1.0 + (ratio - 1.0) * np.clip((ang - radius_deg) / radius_deg, 0.0, 1.0)
        keep = rng.uniform(size=cand.shape[0]) < 1.0 / spacing ** 2
        pts = np.concatenate([pts, cand[keep]])
    pts = pts[:n_cells]
    # two Lloyd-like relaxations towards the region centroids make the cells less ragged (still far from uniform)
    for _ in range(2):
        sv = SphericalVoronoi(pts, radius=1.0, center=np.zeros(3))
        cen = np.stack([sv.vertices[r].mean(axis=0) for r in sv.regions])
        pts = cen / np.linalg.norm(cen, axis=1, keepdims=True)
    sv = SphericalVoronoi(pts, radius=1.0, center=np.zeros(3))
    sv.sort_vertices_of_regions()
    me = max(len(r) for r in sv.regions)
    voc = np.zeros((n_cells, me), np.int32)
    for i, r in enumerate(sv.regions):
        r = np.asarray(r, np.int32)
        a, b, cc = sv.vertices[r[0]], sv.vertices[r[1]], sv.vertices[r[2]]
        if np.dot(pts[i], np.cross(b - a, cc - a)) < 0:
            r = r[::-1]
        voc[i, :len(r)] = r + 1
    latc, lonc = _xyz_to_latlon_rad(pts)
    latv, lonv = _xyz_to_latlon_rad(sv.vertices)
    return MpasMesh(latc, lonc, latv, lonv, voc)


def regional_hex_mesh(proj, x0, y0, q_cells, r_cells, spacing, seed=SEED, jitter=0.12):
    """Perturbed hexagonal mesh in the index plane of Lambert projection `proj`.

    Cell (r, q) sits near x = x0 + (q + 0.5*(r%2))*spacing, y = y0 + r*spacing*sqrt(3)/2 (grid-index
    units of `proj`); cell id = r*q_cells + q.  A ghost ring supplies the rim vertices so that every
    kept cell has a complete 6-vertex polygon; vertices touched by < 3 kept cells exist but give no
    dual triangle (regional rim).  Vertices are spherical circumcentres of the
    lattice triangles, so polygons are the true Voronoi cells while the lattice stays Delaunay.
    """
    rng = np.random.default_rng(seed)
    Q, R = q_cells + 2, r_cells + 2  # ghost lattice
    rr, qq = np.meshgrid(np.arange(R), np.arange(Q), indexing="ij")
    x = x0 + (qq - 1 + 0.5 * ((rr - 1) % 2)) * spacing
    y = y0 + (rr - 1) * spacing * (3.0 ** 0.5 / 2.0)
    x = x + jitter * spacing * rng.uniform(-1.0, 1.0, x.shape)
    y = y + jitter * spacing * rng.uniform(-1.0, 1.0, y.shape)
    lat_deg, lon_deg = proj.ij_to_latlon(x, y)
    lat = np.deg2rad(lat_deg).ravel()
    lon = np.deg2rad(lon_deg).ravel()
    P = latlon_rad_to_xyz(lat, lon)  # ghost lattice points [R*Q][3]
    gid = (rr * Q + qq)

    # triangles between lattice rows g and g+1 (ghost indices); parity refers to the KEPT row index
    # r = g-1, whose odd rows are shifted by +0.5 -> ghost row g is shifted iff (g-1)%2==1.
    g = np.arange(R - 1)[:, None]
    shifted = ((g - 1) % 2 == 1)  # row g shifted
    p00, p01 = gid[:-1, :-1], gid[:-1, 1:]  # (g,q), (g,q+1)
    p10, p11 = gid[1:, :-1], gid[1:, 1:]    # (g+1,q), (g+1,q+1)
    # row g not shifted:  up = (g,q),(g,q+1),(g+1,q);    down = (g,q+1),(g+1,q+1),(g+1,q)
    # row g shifted:      up = (g,q),(g,q+1),(g+1,q+1);  down = (g,q),(g+1,q+1),(g+1,q)
    up = np.stack([p00, p01, np.where(shifted, p11, p10)], axis=-1)
    dn = np.stack([np.where(shifted, p00, p01), p11, p10], axis=-1)
    tris = np.stack([up, dn], axis=2).reshape(-1, 3)  # tid = ((g*(Q-1))+q)*2 + type
    A, B, C = P[tris[:, 0]], P[tris[:, 1]], P[tris[:, 2]]
    n = np.cross(B - A, C - A)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n *= np.sign(np.einsum("ij,ij->i", n, A))[:, None]

    def tid(gg, q_, typ):
        return ((gg * (Q - 1)) + q_) * 2 + typ

    # six triangles around kept cell (ghost coords g=r+1, q=qk+1), CCW from the E-NE one
    gk, qk = np.meshgrid(np.arange(1, R - 1), np.arange(1, Q - 1), indexing="ij")
    sh = ((gk - 1) % 2 == 1)
    ring = np.stack([
        tid(gk, qk, 0),
        np.where(sh, tid(gk, qk, 1), tid(gk, qk - 1, 1)),
        tid(gk, qk - 1, 0),
        tid(gk - 1, qk - 1, 1),
        np.where(sh, tid(gk - 1, qk, 0), tid(gk - 1, qk - 1, 0)),
        tid(gk - 1, qk, 1),
    ], axis=-1).reshape(-1, 6)
    used = np.zeros(tris.shape[0], bool)
    used[ring.ravel()] = True
    newid = np.cumsum(used, dtype=np.int64)  # 1-based ids for used triangles
    voc = newid[ring].astype(np.int32)
    latv, lonv = _xyz_to_latlon_rad(n[used])
    keep = gid[1:-1, 1:-1].ravel()
    lonc = np.where(lon[keep] < 0.0, lon[keep] + 2.0 * np.pi, lon[keep])
    return MpasMesh(lat[keep].copy(), lonc, latv, lonv, voc)


def regional_mesh_for_lambert(proj, nx, ny, n_cells, margin=0.05, seed=SEED):
    """~n_cells-cell hex mesh covering the (nx x ny)-point Lambert domain of `proj` plus `margin`."""
    w, h = (nx - 1) * (1 + 2 * margin), (ny - 1) * (1 + 2 * margin)
    # cell area = spacing^2*sqrt(3)/2
    spacing = (w * h / (n_cells * (3.0 ** 0.5 / 2.0))) ** 0.5
    q_cells = int(np.ceil(w / spacing)) + 1
    r_cells = int(np.ceil(n_cells / q_cells))
    x0 = 1.0 - margin * (nx - 1) - 0.25 * spacing
    y0 = 1.0 - margin * (ny - 1) - ((r_cells - 1) * spacing * (3.0 ** 0.5 / 2.0) - h) / 2.0
    return regional_hex_mesh(proj, x0, y0, q_cells, r_cells, spacing, seed=seed)


def renumber_cells(m, perm):
    """Renumber the cells of a mesh: new cell i is old cell perm[i] (vertices keep their numbers; verticesOnCell refers to
    vertices only, so the connectivity needs no translation)."""
    perm = np.asarray(perm)
    return MpasMesh(m.latCell[perm], m.lonCell[perm], m.latVertex, m.lonVertex, m.verticesOnCell[perm])


def morton_cells(m, bits=12):
    """Cells renumbered along a 2-D Morton (Z-order) curve over the mesh's own lat / lon extent: the locality-preserving
    but NOT row-banded numbering that graph partitioners and space-filling-curve reorderings give real MPAS meshes.
    Consecutive ids are spatial neighbours only inside blocks of 4, 16, 64 ... cells; a row of target points crosses
    many such blocks."""
    lon = np.where(m.lonCell > np.pi, m.lonCell - 2.0 * np.pi, m.lonCell)
    qx = ((lon - lon.min()) / max(np.ptp(lon), 1e-300) * ((1 << bits) - 1)).astype(np.int64)
    qy = ((m.latCell - m.latCell.min()) / max(np.ptp(m.latCell), 1e-300) * ((1 << bits) - 1)).astype(np.int64)
    code = np.zeros(m.nCells, np.int64)
    for b in range(bits):
        code |= ((qx >> b) & 1) << (2 * b)
        code |= ((qy >> b) & 1) << (2 * b + 1)
    return renumber_cells(m, np.argsort(code, kind="stable"))


def cutout_cells(m, bits=21):
    """Cells renumbered as a LIMITED-AREA CUT-OUT of a global mesh numbers them: a regional MPAS mesh is made by cutting cells out of a
    global parent (MPAS-Limited-Area), and the cut-out keeps the parent's relative order.  The parent here is numbered along a 3-D
    Morton curve over the whole unit sphere (as icosahedral_mesh / geodesic_mesh number theirs): the regional cells inherit that
    GLOBAL curve's order -- its cube octants, not the region's own extent, decide where the numbering jumps (morton_cells fits a 2-D
    curve to the region itself: the friendlier case)."""
    xyz = latlon_rad_to_xyz(m.latCell, m.lonCell)
    return renumber_cells(m, _morton_order(xyz, bits=bits))


def shuffle_cells(m, seed=SEED, block=1):
    """Renumber the cells of a mesh: random permutation of blocks of `block` consecutive cells (block=1: fully random).
    Real MPAS meshes are not necessarily numbered along rows; this is the worst case for gather locality."""
    rng = np.random.default_rng(seed + 7)
    nb = (m.nCells + block - 1) // block
    order = rng.permutation(nb)
    perm = np.concatenate([np.arange(b * block, min((b + 1) * block, m.nCells)) for b in order])  # new id i holds old cell perm[i]
    return MpasMesh(m.latCell[perm], m.lonCell[perm], m.latVertex, m.lonVertex, m.verticesOnCell[perm])


def analytic_field(lat_rad, lon_rad, nlev, seed=SEED, cell_fast=True, dtype=np.float64):
    """f(lat,lon,k) = a_k + b_k x + c_k y + d_k z + 0.1 sin(5 lon) cos(3 lat)  (SURVEY s8(d)).
    Returns [nlev][n] (cell-fastest, as the reference holds fields, input_data.F90:653-655) or
    [n][nlev] (level-fastest, MPAS file order)."""
    rng = np.random.default_rng(seed)
    co = rng.uniform(-1.0, 1.0, (nlev, 4))
    xyz = latlon_rad_to_xyz(lat_rad, lon_rad)
    wig = 0.1 * np.sin(5.0 * lon_rad) * np.cos(3.0 * lat_rad)
    f = co[:, :1] + co[:, 1:] @ xyz.T + wig[None, :]
    f = f.astype(dtype)
    return np.ascontiguousarray(f if cell_fast else f.T)


def category_field(n, nlev=1, seed=SEED, ncat=20):
    """Integer-valued float64 field floor(1 + 19u) (land-use-like), [nlev][n]."""
    rng = np.random.default_rng(seed + 1)
    return np.floor(1.0 + (ncat - 1) * rng.uniform(0.0, 1.0, (nlev, n)))


def snow_field(lat_rad, lon_rad, lat0=np.deg2rad(45.0), lon0=np.deg2rad(-100.0 % 360.0)):
    """Non-negative smooth bump (snow-like), [1][n]."""
    x = latlon_rad_to_xyz(lat_rad, lon_rad)
    c = latlon_rad_to_xyz(np.array(lat0), np.array(lon0))
    d2 = np.sum((x - c) ** 2, axis=-1)
    return np.exp(-d2 / 0.02)[None, :]
