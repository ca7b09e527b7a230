"""Shared test helpers: small seeded scenes, ray sets, image metrics."""
import numpy as np
from kkdirect import Scene, look_at, abi, compute_vertex_normals


def rel_mse(img, ref, eps=1e-3):
    """relMSE as BASELINE.json uses it: mean over pixels and channels of (a-b)^2 / (b^2 + eps)."""
    img = np.asarray(img, np.float64); ref = np.asarray(ref, np.float64)
    return float(((img - ref) ** 2 / (ref ** 2 + eps)).mean())


MEASURED = {}      # same-sample-stream relMSE per test, written to gpurun_out/parity_same_stream.json when the session ends (conftest.py)


def record_parity(value, tol):
    """keeps the measured relMSE of the running test beside its bound, so that the bounds can be held against measurements"""
    import os
    name = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    MEASURED.setdefault(name, []).append({"relmse": float(value), "bound": float(tol)})


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


def explain_mismatches(scene, rays, a, b, any_hit=False):
    """Every ray on which two tracers report different primitives is re-evaluated in float64 against the (at most two)
    triangles in dispute and must be shown to be a TIE there -- no class is admitted on the word of the tracers' own t:
      * edge / vertex tie: in float64 the ray passes a reported triangle's edge within the single-precision resolution of
        the reference's projection test (barycentric distance <= bary_tol, see below);
      * interval tie: the float64 hit distance of a disputed triangle lies within t_err of the ray's own (adaptive) mint
        or of maxt -- this is what replaces the "epsilon band" of round 1: rays that start on a surface and meet it again
        at exactly their epsilon;
      * equal-distance tie: both triangles are decisively hit in float64 and their distances differ by <= the sum of
        their t_err (coplanar / duplicate / crossing triangles).
    t_err bounds the rounding of the reference's t = (n_d - o_u n_u - o_v n_v - o_k) / (d_u n_u + d_v n_v + d_k) (TriAccel,
    SURVEY 9.1): 16 ulp of the largest term of the numerator over |denominator|; bary_tol = 64 ulp * coordinate scale /
    shortest edge (>= 1e-5), plus the barycentric shift t_err causes.
    Returns (n_mismatch, n_unexplained, classes) with classes = counts per tie class."""
    pos, idx, _, _ = scene.arrays()
    pos = pos.astype(np.float64)
    eps32 = 2.0 ** -24
    classes = {"edge_vertex": 0, "interval": 0, "equal_distance": 0, "unexplained": 0}
    mism = np.nonzero(a["prim"] != b["prim"])[0]
    for i in mism:
        o = rays[i, 0:3].astype(np.float64); d = rays[i, 4:7].astype(np.float64)
        mint = float(rays[i, 3]); maxt = float(rays[i, 7])
        if np.float32(mint) == np.float32(1e-4):                     # skdtree.cpp:126-129 / :213-214
            m = np.abs(rays[i, 0:3]).max()
            mint = float(np.float32(mint) * np.float32(m if any_hit else max(m, np.float32(1e-4))))
        verdicts = []
        for h in (a[i], b[i]):
            if h["prim"] == 0xFFFFFFFF:
                continue
            p0, p1, p2 = (pos[idx[h["prim"], k]] for k in range(3))
            e1, e2 = p1 - p0, p2 - p0
            n = np.cross(e1, e2)
            den = n @ d
            if abs(den) < 1e-300:
                verdicts.append(("marginal", "edge_vertex", 0.0, 0.0)); continue
            t = (n @ (p0 - o)) / den
            hp = o + t * d
            # barycentrics of the exact hit point
            area2 = n @ n
            u = (np.cross(hp - p0, e2) @ n) / area2
            v = (np.cross(e1, hp - p0) @ n) / area2
            w = 1.0 - u - v
            k = int(np.argmax(np.abs(n)))
            num_terms = max(abs(n @ p0), *(abs(n[j] * o[j]) for j in range(3))) / abs(n[k])
            t_err = 16 * eps32 * num_terms / max(abs(den / n[k]), 1e-300)
            scale = max(np.abs(p0).max(), np.abs(p1).max(), np.abs(p2).max(), np.abs(o).max())
            shortest = min(np.linalg.norm(e1), np.linalg.norm(e2), np.linalg.norm(p2 - p1))
            bary_tol = max(1e-5, 64 * 6e-8 * scale / max(shortest, 1e-300)) + t_err * np.linalg.norm(d) / max(shortest, 1e-300)
            mb = min(u, v, w)
            if abs(mb) <= bary_tol:
                verdicts.append(("marginal", "edge_vertex", t, t_err))
            elif abs(t - mint) <= t_err or (np.isfinite(maxt) and abs(t - maxt) <= t_err):
                verdicts.append(("marginal", "interval", t, t_err))
            elif mb > 0 and mint < t < maxt:
                verdicts.append(("hit", None, t, t_err))
            else:
                verdicts.append(("miss", None, t, t_err))
        cls = None
        marg = [v for v in verdicts if v[0] == "marginal"]
        if marg:
            cls = marg[0][1]
        elif len(verdicts) == 2 and verdicts[0][0] == "hit" and verdicts[1][0] == "hit" and \
                (any_hit or abs(verdicts[0][2] - verdicts[1][2]) <= verdicts[0][3] + verdicts[1][3]):
            cls = "equal_distance"      # (occlusion queries may report any hit along the ray)
        classes[cls or "unexplained"] += 1
    return len(mism), classes["unexplained"], classes
