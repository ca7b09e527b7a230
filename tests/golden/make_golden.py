"""Generates the golden image fixtures from the oracle (the reference itself cannot be built here: SURVEY 8c).
    python tests/golden/make_golden.py
cornell_64x64_4096spp_oracle.npy: developed RGB (float32) of the Cornell box config at 64x64, 4096 spp, seed 1234."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd"))
import kkoracle
from kkdirect import scenes

if __name__ == "__main__":
    film, st = kkoracle.Oracle(scenes.cornell_box(64, 64)).render(4096, seed=1234)
    img = np.where(film[..., 4:5] > 0, film[..., :3] / np.maximum(film[..., 4:5], 1e-30), 0).astype(np.float32)
    np.save(os.path.join(ROOT, "tests", "golden", "cornell_64x64_4096spp_oracle.npy"), img)
    print("samples", st.samples, "mean", img.mean((0, 1)))
