"""Generates the golden image fixtures from the oracle (the reference itself cannot be built here: SURVEY 8c).
    python tests/golden/make_golden.py [name ...]
cornell_64x64_4096spp_oracle.npy: developed RGB (float32) of the Cornell box config at 64x64, 4096 spp, seed 1234.
<name>_<spp>spp_oracle.npy for the config-shaped scenes of tests/golden_scenes.py (every BSDF and emitter; C4-shaped: GGX conductor +
dielectric under the HDR environment map with its 1e4 suns; C3-shaped: 10 000 area emitters): developed RGB at 16384 spp.
The GPU renders the same scenes with the same sample count and ANOTHER seed and must come within relMSE 1e-3
(tests/test_gpu_converged.py, tests/test_gpu_parity.py::test_cornell_converged_reference): the north star's second bar."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "ray-tracer_amd"), os.path.join(ROOT, "tests")]
import kkoracle
from kkdirect import scenes
import golden_scenes


def developed(film):
    return np.where(film[..., 4:5] > 0, film[..., :3] / np.maximum(film[..., 4:5], 1e-30), 0).astype(np.float32)


if __name__ == "__main__":
    want = sys.argv[1:]
    out = os.path.join(ROOT, "tests", "golden")
    if not want or "cornell" in want:
        film, st = kkoracle.Oracle(scenes.cornell_box(64, 64)).render(4096, seed=1234)
        img = developed(film)
        np.save(os.path.join(out, "cornell_64x64_4096spp_oracle.npy"), img)
        print("cornell: samples", st.samples, "mean", img.mean((0, 1)))
    for name, (factory, spp, seed) in golden_scenes.GOLDENS.items():
        if want and name not in want:
            continue
        t = time.time()
        film, st = kkoracle.Oracle(factory()).render(spp, seed=seed)
        img = developed(film)
        np.save(os.path.join(out, "%s_%dspp_oracle.npy" % (name, spp)), img)
        print("%s: samples %d, %.0f s, mean %s, max %.3g" % (name, st.samples, time.time() - t, img.mean((0, 1)), img.max()), flush=True)
