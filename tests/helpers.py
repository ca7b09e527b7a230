"""Shared test helpers: small seeded scenes, ray sets, image metrics."""
import numpy as np
from kkdirect import Scene, look_at, abi, compute_vertex_normals


def rel_mse(img, ref, eps=1e-3):
    """relMSE as BASELINE.json uses it: mean over pixels and channels of (a-b)^2 / (b^2 + eps)."""
    img = np.asarray(img, np.float64); ref = np.asarray(ref, np.float64)
    return float(((img - ref) ** 2 / (ref ** 2 + eps)).mean())


def develop(film):
    w = film[..., 4:5]
    return np.where(w > 0, film[..., :3] / np.maximum(w, 1e-30), 0.0)


def triangle_soup(n_tris=2000, seed=0, extent=4.0, size=0.6, normals=False, n_shapes=3, width=64, height=64):
    """Random triangles in a cube, split over n_shapes diffuse shapes, one emissive quad above."""
    rng = np.random.RandomState(seed)
    sc = Scene()
    per = n_tris // n_shapes
    for s in range(n_shapes):
        c = rng.uniform(-extent, extent, (per, 1, 3))
        p = (c + rng.uniform(-size, size, (per, 3, 3))).reshape(-1, 3).astype(np.float32)
        idx = np.arange(3 * per, dtype=np.uint32).reshape(-1, 3)
        m = sc.twosided(sc.diffuse(rng.uniform(0.2, 0.8, 3)))
        sc.add_mesh(p, idx, m, normals=compute_vertex_normals(p, idx) if normals else None)
    q = np.array([[-2, extent + 1, -2], [-2, extent + 1, 2], [2, extent + 1, 2], [2, extent + 1, -2]], np.float32)
    sc.add_mesh(q, [[0, 2, 1], [0, 3, 2]], sc.diffuse((0, 0, 0)), radiance=(20, 18, 15))
    sc.set_sensor(look_at((0, 0, -3 * extent), (0, 0, 0), (0, 1, 0)), 50.0, width, height, near=0.1, far=100.0)
    return sc


def chord_rays(n, center, radius, seed=0, mint=0.0):
    """kdbench protocol (reference src/utils/kdbench.cpp:223-233): chords between two uniform points on a sphere."""
    rng = np.random.RandomState(seed)
    def sph(k):
        v = rng.normal(size=(k, 3)); return v / np.linalg.norm(v, axis=1, keepdims=True)
    p1 = np.asarray(center) + radius * sph(n); p2 = np.asarray(center) + radius * sph(n)
    d = p2 - p1; d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = p1; rays[:, 3] = mint; rays[:, 4:7] = d; rays[:, 7] = np.inf
    return rays


def classify_mismatches(scene, rays, a, b):
    """For rays whose primitive ids differ between two tracers, decide whether each one is a documented tie:
      (1) edge/vertex tie: the ray passes within the single-precision resolution of the reference's projection test of
          an edge of either reported triangle (barycentric distance < max(1e-5, 64 ulp * |coordinates| / shortest edge));
      (2) equal-distance tie: both hits at the same t (coplanar / duplicate / crossing triangles);
      (3) epsilon-band tie: one of the hits lies within 64x the ray's own adaptive epsilon of its origin -- a ray that
          starts ON a surface, where the plane equation of the reference's TriAccel test (n_d - o.n)/(d.n) has cancelled
          to noise and reports (or misses) the surface it starts on.
    Returns the number of mismatches in none of these classes."""
    pos, idx, _, _ = scene.arrays()
    bad = 0
    for i in np.nonzero(a["prim"] != b["prim"])[0]:
        ok = False
        ta, tb = float(a["t"][i]), float(b["t"][i])
        if np.isfinite(ta) and np.isfinite(tb) and abs(ta - tb) <= 1e-4 * max(abs(ta), 1e-6):
            ok = True
        o = rays[i, 0:3].astype(np.float64); d = rays[i, 4:7].astype(np.float64)
        mint = float(rays[i, 3])
        if np.float32(mint) == np.float32(1e-4):
            mint *= max(np.abs(o).max(), 1e-4)
        if min(ta, tb) <= 64 * mint and mint > 0:
            ok = True
        for h in (a[i], b[i]):
            if h["prim"] == 0xFFFFFFFF:
                continue
            p0, p1, p2 = (pos[idx[h["prim"], k]].astype(np.float64) for k in range(3))
            e1, e2 = p1 - p0, p2 - p0
            pv = np.cross(d, e2); det = e1 @ pv
            if abs(det) < 1e-30:
                ok = True; continue
            tv = o - p0; u = (tv @ pv) / det; qv = np.cross(tv, e1); v = (d @ qv) / det
            scale = max(np.abs(p0).max(), np.abs(p1).max(), np.abs(p2).max(), np.abs(o).max())
            shortest = min(np.linalg.norm(e1), np.linalg.norm(e2), np.linalg.norm(p2 - p1))
            tol = max(1e-5, 64 * 6e-8 * scale / max(shortest, 1e-30))
            if min(u, v, 1 - u - v) < tol:
                ok = True
        if not ok:
            bad += 1
    return bad
