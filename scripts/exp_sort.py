"""Experiment: how much faster does the closest-hit kernel trace C2-like BSDF rays when they are sorted?
Rays: origins = camera hit points of C2 at 480x270, 64 cosine-hemisphere directions each (8.3 M rays)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd"))
import numpy as np, torch
import kkdirect
from kkdirect import scenes
W, H, G = 480, 270, 64
sc = scenes.sponza_class(W, H)
ctx = kkdirect.Context(device=0, seed=1).upload(sc)
eye = np.array((1.5, 4.2, 7.6)); tgt = np.array((16.0, 4.6, 6.4)); up = np.array((0., 1, 0))
f = tgt - eye; f /= np.linalg.norm(f); r = np.cross(up, f); r /= np.linalg.norm(r); u = np.cross(f, r)
# pixels in Morton order inside 32x32 tiles (like the renderer)
ys, xs = np.mgrid[0:H, 0:W]
def morton(x, y):
    k = np.zeros_like(x)
    for b in range(5): k |= ((x >> b) & 1) << (2 * b) | ((y >> b) & 1) << (2 * b + 1)
    return k
tile = (ys // 32) * ((W + 31) // 32) + xs // 32
order = np.lexsort((morton(xs % 32, ys % 32).ravel(), tile.ravel()))
px, py, ptile = xs.ravel()[order], ys.ravel()[order], tile.ravel()[order]
t = np.tan(np.radians(62.0) / 2)
dx = (2 * (px + 0.5) / W - 1) * t; dy = (1 - 2 * (py + 0.5) / H) * t * H / W
d = f[None] + dx[:, None] * (-r[None]) + dy[:, None] * u[None]
d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros((len(d), 8), np.float32); rays[:, :3] = eye; rays[:, 3] = 0; rays[:, 4:7] = d; rays[:, 7] = np.inf
hits = ctx.trace(rays)
ok = np.isfinite(hits["t"])
print("camera hits", ok.mean())
P = eye[None] + d * np.where(ok, hits["t"], 1.0)[:, None]
N = -d   # proxy normal
rng = np.random.default_rng(1)
n = len(P) * G
u1 = rng.random(n, np.float32); u2 = rng.random(n, np.float32)
rr = np.sqrt(u1); ph = 2 * np.pi * u2
lx, ly, lz = rr * np.cos(ph), rr * np.sin(ph), np.sqrt(np.maximum(0, 1 - u1))
Nn = np.repeat(N, G, axis=0)
a = np.where(np.abs(Nn[:, :1]) > 0.9, np.array([[0., 1, 0]]), np.array([[1., 0, 0]]))
T = np.cross(a, Nn); T /= np.linalg.norm(T, axis=1, keepdims=True); B = np.cross(Nn, T)
D = (T * lx[:, None] + B * ly[:, None] + Nn * lz[:, None]).astype(np.float32)
O = np.repeat(P, G, axis=0).astype(np.float32)
R = np.zeros((n, 8), np.float32); R[:, :3] = O; R[:, 3] = 1e-4; R[:, 4:7] = D; R[:, 7] = np.inf
R[~np.repeat(ok, G), 7] = -1
tl = np.repeat(ptile, G)
# direction cell: dominant axis/sign (6) x 4x4 grid on the face
ax = np.argmax(np.abs(D), axis=1); sg = (np.take_along_axis(D, ax[:, None], 1)[:, 0] < 0)
oth = np.stack([np.where(ax == 0, D[:, 1], D[:, 0]), np.where(ax == 2, D[:, 1], D[:, 2])], 1) / np.abs(np.take_along_axis(D, ax[:, None], 1))
def cell(k): 
    g = np.clip(((oth + 1) * 0.5 * k).astype(np.int64), 0, k - 1)
    return (ax * 2 + sg) * k * k + g[:, 0] * k + g[:, 1]
idx = np.arange(n)
variants = {"natural (pixel-major)": idx,
            "tile, then 6x2x2 direction cell": np.lexsort((idx, cell(2), tl)),
            "tile, then 6x4x4 direction cell": np.lexsort((idx, cell(4), tl)),
            "tile, then 6x8x8 direction cell": np.lexsort((idx, cell(8), tl)),
            "global 6x4x4 cell, then tile": np.lexsort((idx, tl, cell(4))),
            "random shuffle": rng.permutation(n)}
out = torch.zeros(n * 4, dtype=torch.float32, device="cuda")
for name, perm in variants.items():
    dev = torch.from_numpy(np.ascontiguousarray(R[perm])).cuda()
    for any_hit in (False, True):
        ctx.trace_device(dev.data_ptr(), n, out.data_ptr(), any_hit)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): ctx.trace_device(dev.data_ptr(), n, out.data_ptr(), any_hit)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("%-36s any=%d  %.3f ms  %.0f Mrays/s" % (name, any_hit, dt * 1e3, n / dt / 1e6), flush=True)
