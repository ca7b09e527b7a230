import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ray-tracer_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, kkdirect, kkoracle, helpers
from kkdirect import abi, scenes
from test_gpu_parity import _material_scene
sc = _material_scene()
o = kkoracle.Oracle(sc)
for kw in (dict(max_depth=12, rr_depth=5), dict(max_depth=30, rr_depth=5), dict(max_depth=64, rr_depth=5), dict(max_depth=-1, rr_depth=5)):
    ref, so = o.render(32, seed=1, integrator=abi.INTEGRATOR_PATH, **kw)
    with kkdirect.Context(device=0, seed=1).upload(sc) as ctx:
        film, sg = ctx.render(32, integrator=abi.INTEGRATOR_PATH, **kw)
    a, b = helpers.develop(film), helpers.develop(ref)
    d = np.abs(a - b).sum(-1)
    print(kw, "relMSE %.3g" % helpers.rel_mse(a, b), "mean ratio %.5f" % (a.mean() / b.mean()), "rays", sg.rays_closest, so.rays_closest, sg.rays_shadow, so.rays_shadow,
          "pixels differing >1e-3:", int((d > 1e-3).sum()), flush=True)
