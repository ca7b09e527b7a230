"""Multi-device rendering inside the library (kkd_multi_*, include/kkdirect.h), on the one GPU of the test box: the device
is listed twice, so two contexts (each with its own streams and buffers) render their shares side by side and the merge
code -- tile gather with borders, whole-film copy + add, the RCCL reduce with one rank -- runs for real.  The merged film
must equal the film of a single context: the sample stream is keyed by pixel and sample index, not by rank."""
import numpy as np
import pytest
import torch   # before libkkdirect is loaded (see test_zz_no_alloc.py)
from kkdirect import abi, scenes
from test_gpu_parity import _material_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
@pytest.mark.parametrize("split", ["tile", "sample"])
def test_listed_twice_equals_single_context(gpu, split, devices):
    sc = _material_scene(200, 136)                   # 7 x 5 tiles of 32: partial tiles on two sides
    with gpu.Context(device=0, seed=4).upload(sc) as ctx:
        full, st = ctx.render(12)
    with gpu.MultiContext(devices, split=split, seed=4).upload(sc) as m:
        film, sm = m.render(12)
        assert m.merge_path == ("tile-gather" if split == "tile" else "copy-add")
        assert sm.samples == st.samples and sm.class_rays[abi.RAYS_PRIMARY] == st.class_rays[abi.RAYS_PRIMARY]
        np.testing.assert_allclose(film, full, rtol=1e-4, atol=1e-5)
        # a second frame through the same handle (buffers reused), other block size and high-quality edges off/on
        film2, _ = m.render(12, block_size=16)
        np.testing.assert_allclose(film2, full, rtol=1e-4, atol=1e-5)
        # the caller's own shard is refined, not replaced: two "nodes" of this job sum to the whole
        if split == "tile":
            parts = [m.render(12, tile_index=k, tile_count=2)[0] for k in range(2)]
        else:
            parts = [m.render(6, sample_first=6 * k)[0] for k in range(2)]
        np.testing.assert_allclose(parts[0] + parts[1], full, rtol=1e-4, atol=1e-5)


def test_high_quality_edges_through_the_tile_gather(gpu):
    sc = scenes.cornell_box(100, 70)
    sc.sensor.film_high_quality_edges = 1
    with gpu.Context(device=0, seed=2).upload(sc) as ctx:
        full, _ = ctx.render(8)
    with gpu.MultiContext([0, 0], split="tile", seed=2).upload(sc) as m:
        film, _ = m.render(8)
    np.testing.assert_allclose(film, full, rtol=1e-4, atol=1e-5)


def test_one_device_and_device_film(gpu):
    """n = 1: tile mode is the plain render; sample mode goes through the RCCL reduce with one rank when librccl loads
    (the plumbing the 8-GPU run uses), and the device-film entry point leaves the result on devices[0]"""
    sc = scenes.cornell_box(64, 64)
    with gpu.Context(device=0, seed=3).upload(sc) as ctx:
        full, _ = ctx.render(8)
    for split in ("tile", "sample"):
        with gpu.MultiContext([0], split=split, seed=3).upload(sc) as m:
            t = torch.full((64 * 64 * 5,), 3.0, dtype=torch.float32, device="cuda")
            st = m.render_device(t.data_ptr(), 8)
            torch.cuda.synchronize()
            assert m.merge_path in (("single",) if split == "tile" else ("single", "rccl-reduce"))
            np.testing.assert_allclose(t.cpu().numpy().reshape(64, 64, 5), full, rtol=1e-4, atol=1e-5)
            assert st.samples == 64 * 64 * 8


def test_multi_error_behaviour(gpu):
    with pytest.raises(gpu.KKDError):
        gpu.MultiContext([0, 99])
    sc = scenes.cornell_box(32, 32)
    with gpu.MultiContext([0, 0], split="sample") as m:
        with pytest.raises(gpu.KKDError, match="no scene"):
            m.render(4)
        m.upload(sc)
        with pytest.raises(gpu.KKDError, match="spp >= number of devices"):
            m.render(1)
        with pytest.raises(gpu.KKDError):
            m.render(4, bsdf_samples=0)
        seen = []
        m.set_progress_callback(lambda done, total: seen.append((done, total)))
        film, st = m.render(4)
        assert st.samples == 32 * 32 * 4 and film[..., 4].sum() > 0
        assert seen and seen[-1][0] == seen[-1][1] >= st.samples and all(a[0] <= b[0] for a, b in zip(seen, seen[1:]))


@pytest.mark.parametrize("sampler", [abi.SAMPLER_LDSAMPLER, abi.SAMPLER_STRATIFIED, abi.SAMPLER_HALTON, abi.SAMPLER_HAMMERSLEY])
@pytest.mark.parametrize("split", ["tile", "sample"])
def test_samplers_share_one_point_set_across_ranks(gpu, split, sampler):
    """sample-split hands every rank a slice of each pixel's one point set (sample_count_total is filled in by kkd_multi_render), tile
    mode its own pixels: the merged film is the single-context film for every sampler"""
    sc = scenes.cornell_box(96, 80)
    with gpu.Context(device=0, seed=6).upload(sc) as ctx:
        full, _ = ctx.render(16, sampler=sampler)
    with gpu.MultiContext([0, 0], split=split, seed=6).upload(sc) as m:
        film, _ = m.render(16, sampler=sampler)
    np.testing.assert_allclose(film, full, rtol=1e-4, atol=1e-5)


def test_more_ranks_than_tiles_after_a_normal_render(gpu):
    """a rank that gets no tile in THIS render must not pack the tiles an earlier render on the same handle left in its list
    (round-2 review: multiRender took the count from the context's stale list)"""
    sc = scenes.cornell_box(100, 70)
    with gpu.Context(device=0, seed=8).upload(sc) as ctx:
        full, st = ctx.render(8)
    with gpu.MultiContext([0, 0, 0], split="tile", seed=8).upload(sc) as m:
        a, _ = m.render(8)                               # 4 x 3 tiles of 32: every rank has four
        np.testing.assert_allclose(a, full, rtol=1e-4, atol=1e-5)
        b, sb = m.render(8, block_size=128)              # ONE tile: ranks 1 and 2 have none
        assert sb.samples == st.samples
        np.testing.assert_allclose(b, full, rtol=1e-4, atol=1e-5)
        c, _ = m.render(8, block_size=64)                # 2 x 2 tiles: rank 0 two, ranks 1 and 2 one each
        np.testing.assert_allclose(c, full, rtol=1e-4, atol=1e-5)
        d, _ = m.render(8)
        np.testing.assert_allclose(d, full, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("split", ["tile", "sample"])
def test_flush_through_the_multi_handle(gpu, split):
    """`mitsuba -r`: kkd_multi_request_flush during a render hands out the sum of the devices' films so far (mitsuba.cpp:105-125
    -> RenderQueue::flush); a request made before any batch has finished is answered, one made when all are queued is answered
    with the finished films"""
    sc = scenes.cornell_box(128, 128)
    flushed, seen = [], []
    with gpu.MultiContext([0, 0], split=split, seed=5, max_samples_in_flight=1 << 14).upload(sc) as m:
        m.set_flush_callback(lambda film, done, total: flushed.append((film, done, total)))

        def progress(done, total):
            seen.append((done, total))
            if len(seen) in (2, 9):
                m.request_flush()
        m.set_progress_callback(progress)
        final, st = m.render(32)
    assert len(seen) > 9 and flushed, (len(seen), len(flushed))
    for film, done, total in flushed:
        assert film.shape == final.shape and np.isfinite(film).all() and film.min() >= 0
        assert 0 < done <= total
        # a partial film holds a subset of the final film's samples: nowhere more (the additions differ in order only)
        assert (film <= final * (1 + 1e-4) + 1e-4).all()
        assert film[..., 4].sum() <= final[..., 4].sum() * (1 + 1e-5)
    first = flushed[0][0]
    assert 0 < first[..., 4].sum() < final[..., 4].sum() * 0.9      # taken while the render was running
    with gpu.Context(device=0, seed=5).upload(sc) as ctx:
        ref, _ = ctx.render(32)
    np.testing.assert_allclose(final, ref, rtol=1e-4, atol=1e-5)
