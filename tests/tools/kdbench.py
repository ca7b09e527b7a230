"""The reference's two ray-throughput protocols on this library (run on the GPU box; prints what profiles/r3_kdbench.txt holds).

  * src/utils/kdbench.cpp:210-242 -- 5 M chords between two uniform points on the bounding sphere of the tree's AABB,
    closest hit (rayIntersect(ray, its)), one thread, best of three, "MRays/s";
  * src/tests/test_kd.cpp:86-131 -- 10 M such chords, any hit (rayIntersect(ray)), three iterations.

GPU: rays resident in device memory, kkd_trace_device (the same kernels the render path's traversal stages use, one launch),
timed with the library's own HIP events (kkd_stats.seconds).  CPU: the oracle's kd-tree (Havran traversal, TriAccel test) on ONE
thread, as kdbench does, on a bounded sample of the same chords.  usage: python tests/tools/kdbench.py [c2|c3|c4] [cpu_rays]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ray-tracer_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np, torch
import kkdirect, kkoracle, helpers
from bench import make_workload

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
cpu_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
scene, _, label = make_workload(wl, 1.0)
lo, hi = scene.aabb()
center = 0.5 * (np.asarray(lo, np.float64) + np.asarray(hi, np.float64))
radius = float(np.linalg.norm(np.asarray(hi, np.float64) - center))           # AABB::getBSphere
print("%s: bounding sphere c = %s r = %.4f" % (label, np.round(center, 4).tolist(), radius), flush=True)
with kkdirect.Context(device=0, seed=1).upload(scene) as ctx:
    for name, n, any_hit in (("kdbench.cpp, closest hit, 5 M chords", 5_000_000, False), ("test_kd.cpp, any hit, 10 M chords", 10_000_000, True)):
        best = 0.0
        for it in range(3):
            rays = torch.from_numpy(helpers.chord_rays(n, center, radius, seed=10 * it + (1 if any_hit else 0))).cuda()
            out = torch.zeros(n * 4, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            st = ctx.trace_device(rays.data_ptr(), n, out.data_ptr(), any_hit=any_hit)
            hits = int((out.view(-1, 4)[:, 0] < float("inf")).sum().item()) if not any_hit else -1
            mrays = n / st.seconds / 1e6
            best = max(best, mrays)
            print("  GPU  %s: iteration %d  %.3f ms  -> %.1f MRays/s%s" % (name, it, st.seconds * 1e3, mrays, "" if any_hit else "  (%d intersections)" % hits), flush=True)
        print("  GPU  %s: best of three %.1f MRays/s" % (name, best), flush=True)
o = kkoracle.Oracle(scene)
for name, any_hit in (("closest hit", False), ("any hit", True)):
    rays = helpers.chord_rays(cpu_rays, center, radius, seed=1 if any_hit else 0)
    t0 = time.perf_counter()
    o.trace(rays, any_hit=any_hit)
    dt = time.perf_counter() - t0
    print("  CPU oracle (kd-tree, %s, %d chords, one thread): %.3f MRays/s" % (name, cpu_rays, cpu_rays / dt / 1e6), flush=True)
