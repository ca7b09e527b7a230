"""Print the same-sample-stream image parity numbers quoted in DESIGN.md / README.md (GPU vs oracle)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ray-tracer_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, kkdirect, kkoracle, helpers
from kkdirect import scenes
from test_gpu_parity import _material_scene
for name, sc, spp in (("cornell 128x128 x64", scenes.cornell_box(128, 128), 64), ("all materials 96x64 x64", _material_scene(), 64),
                      ("sponza-class 20k 256x144 x16", scenes.sponza_class(256, 144, target_tris=20_000), 16)):
    ref, so = kkoracle.Oracle(sc).render(spp, seed=1)
    with kkdirect.Context(device=0, seed=1).upload(sc) as ctx:
        film, sg = ctx.render(spp)
    print(name, "relMSE %.3g" % helpers.rel_mse(helpers.develop(film), helpers.develop(ref)), "closest rays", sg.rays_closest, so.rays_closest, flush=True)
