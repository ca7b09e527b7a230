"""Round-based traversal (kTraceRound) under stress, on the GPU: tiny iteration caps and several rounds (almost every ray is
parked, shards overflow, parked rays are parked again), and stacks that spill into private memory (such lanes must not be
parked).  The caps are read from the environment once per process, so the stress configuration runs in ONE child process."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest
import kkoracle
from kkdirect import Scene, scenes
import helpers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys, os
sys.path[:0] = [os.path.join(%(root)r, "ray-tracer_amd"), os.path.join(%(root)r, "oracle"), os.path.join(%(root)r, "tests")]
import numpy as np, kkdirect, kkoracle, helpers
from kkdirect import scenes
from test_gpu_parity import _material_scene
out = {}
for name, sc, spp, kw in (("cornell", scenes.cornell_box(96, 96), 16, {}), ("materials", _material_scene(64, 48), 16, dict(bsdf_samples=2, light_samples=2)),
                          ("sponza", scenes.sponza_class(96, 54, target_tris=20000), 8, {})):
    ref, so = kkoracle.Oracle(sc).render(spp, seed=1, **kw)
    with kkdirect.Context(device=0, seed=1, max_samples_in_flight=%(batch)d).upload(sc) as ctx:
        film, sg = ctx.render(spp, **kw)
    out[name] = dict(relmse=helpers.rel_mse(helpers.develop(film), helpers.develop(ref)), samples=[int(sg.samples), int(so.samples)],
                     closest=[int(sg.rays_closest), int(so.rays_closest)], finite=bool(np.isfinite(film).all()))
print("RESULT " + json.dumps(out))
"""


@pytest.mark.parametrize("env", [dict(KKD_ITER_CAP_CLOSEST="3", KKD_ITER_CAP_ANY="2", KKD_ITER_CAP_LATER="2", KKD_TRACE_ROUNDS="5"),
                                 dict(KKD_ITER_CAP_CLOSEST="9", KKD_ITER_CAP_ANY="7", KKD_ITER_CAP_LATER="5", KKD_TRACE_ROUNDS="3"),
                                 dict(KKD_ITER_CAP_CLOSEST="0", KKD_ITER_CAP_ANY="0"),
                                 # two rounds (the default): nearly every ray goes through the persistent tail round, more than its buffer holds
                                 dict(KKD_ITER_CAP_CLOSEST="3", KKD_ITER_CAP_ANY="2"),
                                 dict(KKD_ITER_CAP_CLOSEST="6", KKD_ITER_CAP_ANY="4", KKD_TAIL_PERSIST="4"),
                                 dict(KKD_TAIL_PERSIST="0", KKD_TRACE_ROUNDS="3"),                      # the plain resume rounds
                                 dict(KKD_TAIL_PERSIST="0", KKD_ITER_CAP_CLOSEST="4", KKD_ITER_CAP_ANY="3"),
                                 # stray settings must not change a result: the release build ignores the ablation bits of
                                 # $KKD_DEBUG (they exist only with -DKKD_ABLATION) and clamps the performance knobs
                                 dict(KKD_DEBUG="7", KKD_STREAMS="-3", KKD_TRACE_ROUNDS="-1", KKD_ITER_CAP_LATER="-5"),
                                 dict(KKD_DEBUG="71", KKD_STREAMS="99", KKD_TRACE_ROUNDS="1000", KKD_ITER_CAP_CLOSEST="99999")])
def test_rounds_under_stress_give_the_same_image(gpu, env):
    e = dict(os.environ); e.update(env)
    p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, batch=65536)], env=e, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    for name, r in res.items():
        assert r["finite"] and r["samples"][0] == r["samples"][1] and r["closest"][0] == r["closest"][1], (name, r)
        assert r["relmse"] < (1e-5 if name == "cornell" else 2e-4), (name, r)


def test_deep_stacks_spill_and_are_not_parked(gpu):
    """20 000 coincident triangles: the builder can only halve the range, the tree is as deep as it gets and every ray that
    hits the triangle hits every box, so the per-lane stack grows past its 16 LDS words into the private part."""
    n = 20000
    tri = np.array([[-1, 0, -1], [1, 0, -1], [0, 0, 1.5]], np.float32)
    pos = np.tile(tri, (n, 1)); idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    sc = Scene(); sc.add_mesh(pos, idx, sc.diffuse())
    sc.add_mesh([[-3, 2, -3], [-3, 2, 3], [3, 2, 3], [3, 2, -3]], [[0, 2, 1], [0, 3, 2]], sc.diffuse((0, 0, 0)), radiance=(5, 5, 5))
    sc.set_sensor(np.eye(4), 45, 16, 16)
    o = kkoracle.Oracle(sc)
    rng = np.random.RandomState(3)
    rays = np.zeros((2000, 8), np.float32)
    rays[:, 0:3] = rng.uniform(-1.5, 1.5, (2000, 3)) + [0, 3, 0]; rays[:, 1] = 1.0
    tgt = rng.uniform(-1.2, 1.2, (2000, 3)); tgt[:, 1] = 0
    d = tgt - rays[:, 0:3]; rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True); rays[:, 7] = np.inf
    with gpu.Context(device=0).upload(sc) as ctx:
        assert ctx.accel_info()["n_leaf_tris"] == n + 2
        a, b = o.trace(rays), ctx.trace(rays)
        hit_a, hit_b = a["prim"] != 0xFFFFFFFF, b["prim"] != 0xFFFFFFFF
        assert (hit_a != hit_b).mean() < 2e-3                       # rays through the outline of the triangle
        both = hit_a & hit_b
        assert both.sum() > 500
        np.testing.assert_allclose(b["t"][both], a["t"][both], rtol=1e-5)
        assert (b["prim"][both] < n).all() and (a["prim"][both] < n).all()      # some copy of the stacked triangle: equal-distance tie
        sa, sb = o.trace(rays, any_hit=True), ctx.trace(rays, any_hit=True)
        assert ((sa["prim"] != 0xFFFFFFFF) != (sb["prim"] != 0xFFFFFFFF)).mean() < 2e-3
        film, st = ctx.render(4)                                   # and through the round-based render path
        ref, _ = o.render(4, seed=1)
    assert helpers.rel_mse(helpers.develop(film), helpers.develop(ref)) < 1e-4


@pytest.mark.gpu
def test_two_batch_lanes_equal_one_lane():
    """Renders keep two batches in flight on two streams (kkd_api.hip, 'batch lanes'); KKD_FLAG_ONE_LANE runs the same batches
    one after the other.  Same samples, same rays, same film up to the order of the float atomics -- for the library's own
    batch split (an even number of equal batches) and for a caller-chosen odd one."""
    import kkdirect
    from kkdirect import scenes
    sc = scenes.sponza_class(512, 288, target_tris=20_000)
    for in_flight in (0, 700_000):
        films, stats = [], []
        for one in (True, False):
            with kkdirect.Context(device=0, seed=3, max_samples_in_flight=in_flight, one_lane=one).upload(sc) as ctx:
                f, s = ctx.render(32)            # 4.7 M samples: 2 batches by default, 7 with 700 k in flight
                films.append(f); stats.append(s)
        assert stats[0].samples == stats[1].samples == 512 * 288 * 32
        assert stats[0].rays_closest == stats[1].rays_closest and stats[0].rays_shadow == stats[1].rays_shadow
        assert stats[1].stage_launches[0] >= 2          # more than one batch, so the second lane was used
        np.testing.assert_allclose(films[0], films[1], rtol=2e-4, atol=1e-4)
        assert np.isfinite(films[1]).all()


@pytest.mark.gpu
def test_progress_callback_reports_finished_batches():
    """kkd_set_progress_callback: called between batches with the samples whose results are in the film, monotonically, and
    with done == total at the end (the adapter's hook for ProgressReporter / RenderQueue::signalWorkEnd)."""
    import kkdirect
    from kkdirect import scenes
    sc = scenes.cornell_box(128, 128)
    with kkdirect.Context(device=0, seed=1, max_samples_in_flight=65536).upload(sc) as ctx:
        calls = []
        ctx.set_progress_callback(lambda done, total: calls.append((done, total)))
        film, st = ctx.render(32)                     # 524 288 samples: 8 batches
        assert len(calls) >= 2 and all(t == calls[0][1] for _, t in calls)
        dones = [d for d, _ in calls]
        assert dones == sorted(dones) and len(set(dones)) == len(dones)
        assert dones[-1] == calls[-1][1] >= st.samples
        n = len(calls)
        ctx.set_progress_callback(None)
        ctx.render(8)
        assert len(calls) == n


@pytest.mark.gpu
def test_flush_hands_out_the_film_so_far():
    """kkd_request_flush / kkd_set_flush_callback (what `mitsuba -r` does through FlushThread -> Scene::flush -> Film::develop):
    the copy holds exactly the batches that are done -- pixels of finished tiles further than the filter border from unfinished
    ones already have their final value, pixels of unfinished tiles further than the border from finished ones are empty."""
    import kkdirect
    from kkdirect import scenes
    W = H = 128; bs = 32; spp = 32
    sc = scenes.cornell_box(W, H)
    with kkdirect.Context(device=0, seed=1, max_samples_in_flight=65536).upload(sc) as ctx:
        flushes = []
        ctx.set_flush_callback(lambda film, done, total: flushes.append((film, done, total)))
        ctx.set_progress_callback(lambda done, total: ctx.request_flush())      # allowed from the callback, like a cancel request
        final, st = ctx.render(spp)                                             # 524 288 samples: 8 batches of two tiles
        assert len(flushes) >= 2
        tile_of = (np.arange(H)[:, None] // bs) * (W // bs) + np.arange(W)[None, :] // bs     # tiles are rendered row-major
        seen_partial = False
        for film, done, total in flushes:
            assert film.shape == (H, W, 5) and total == W * H * spp and 0 < done <= total and done % (bs * bs * spp) == 0
            complete = tile_of < done // (bs * bs * spp)
            pad = np.pad(complete, 2, mode="edge")
            windows = np.stack([pad[dy:dy + H, dx:dx + W] for dy in range(5) for dx in range(5)])
            inside, outside = windows.all(0), ~windows.any(0)
            np.testing.assert_allclose(film[inside], final[inside], rtol=1e-5, atol=1e-7)
            assert not film[outside].any()
            assert (film <= final * (1 + 1e-5) + 1e-7).all()
            seen_partial |= done < total
        assert seen_partial
        dones = [d for _, d, _ in flushes]
        assert dones == sorted(dones)
        n = len(flushes)
        ctx.set_progress_callback(None)
        ctx.request_flush()                       # a request from before the render is dropped at its start
        again, _ = ctx.render(spp)
        assert len(flushes) == n
        np.testing.assert_allclose(again, final, rtol=1e-4, atol=1e-6)
        ctx.set_flush_callback(None)
