"""Tie rate of hit ids (GPU vs oracle Wald test) on the full C2 scene, camera + secondary rays.
   argv[1] = "proj": GPU leaf test = KKD_FLAG_PROJECTION_TEST instead of the watertight test."""
import sys, os, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'oracle')); sys.path.insert(0, os.path.join(R, 'ray-tracer_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import kkoracle, kkdirect, helpers
from kkdirect import scenes
sc = scenes.sponza_class(1920, 1080)
t = time.time(); o = kkoracle.Oracle(sc); print('oracle build', time.time() - t, flush=True)
proj = len(sys.argv) > 1 and sys.argv[1] == 'proj'
ctx = kkdirect.Context(projection_test=proj).upload(sc)
print('projection test' if proj else 'watertight test', flush=True)
rng = np.random.RandomState(0)
n = 2_000_000
cam = o.camera_rays(rng.rand(n, 2).astype(np.float32) * [1920, 1080])
t = time.time(); a = o.trace(cam); print('oracle trace', time.time() - t, flush=True)
b = ctx.trace(cam)
mis = a['prim'] != b['prim']
print('camera: t bit-equal on equal ids', (a['t'][~mis].view(np.uint32) == b['t'][~mis].view(np.uint32)).mean())
print('camera: mismatches', mis.sum(), 'rate', mis.mean(), 'unexplained', helpers.classify_mismatches(sc, cam, a, b), flush=True)
ok = a['prim'] != 0xFFFFFFFF
p = cam[ok, 0:3] + cam[ok, 4:7] * a['t'][ok, None]
d = rng.normal(size=(len(p), 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
sec = np.zeros((len(p), 8), np.float32); sec[:, 0:3] = p; sec[:, 3] = 1e-4; sec[:, 4:7] = d; sec[:, 7] = np.inf
a2 = o.trace(sec); b2 = ctx.trace(sec)
mis2 = a2['prim'] != b2['prim']
print('secondary: t bit-equal on equal ids', (a2['t'][~mis2].view(np.uint32) == b2['t'][~mis2].view(np.uint32)).mean())
print('secondary: mismatches', mis2.sum(), 'rate', mis2.mean(), 'unexplained', helpers.classify_mismatches(sc, sec, a2, b2), flush=True)
idx = np.nonzero(mis2)[0][:8]
for i in idx: print(sec[i], a2[i], b2[i])
