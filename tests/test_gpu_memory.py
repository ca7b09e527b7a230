"""Device memory pressure: the wavefront buffers of a render are sized by the batch (two lanes of 64 M camera samples by
default, ~28 GB); next to another tenant of the GPU -- here a torch tensor that takes all but a few GB -- the render halves
its batches (down to 1 M samples, then one lane) instead of failing (round-2 review; the reference's CPU path has no
counterpart: its ImageBlocks are per worker thread, librender/renderproc.cpp:26-66)."""
import numpy as np
import pytest
import torch
from kkdirect import scenes

pytestmark = pytest.mark.gpu


def test_render_backs_off_under_memory_pressure(gpu, capfd, monkeypatch):
    monkeypatch.setenv("KKD_VERBOSE", "1")
    sc = scenes.sponza_class(1280, 720, target_tris=60_000)
    with gpu.Context(device=0, seed=3).upload(sc) as ctx:
        ref, st = ctx.render(16)
    torch.cuda.empty_cache()
    with gpu.Context(device=0, seed=3).upload(sc) as ctx:
        free, _ = torch.cuda.mem_get_info()
        keep = 3 << 30                                            # what the render may still take: 3 GiB (it wants ~4 GB and must leave 0.75 GB)
        hog = torch.empty(max(0, free - keep), dtype=torch.uint8, device="cuda")
        try:
            film, s2 = ctx.render(16)
        finally:
            del hog
            torch.cuda.empty_cache()
    assert "backing off" in capfd.readouterr().err           # the render did run out of memory first
    assert s2.samples == st.samples and s2.rays_closest == st.rays_closest and s2.rays_shadow == st.rays_shadow
    np.testing.assert_allclose(film, ref, rtol=1e-4, atol=1e-5)
