"""ldsampler (reference src/samplers/ldsampler.cpp:148-229, Mitsuba's default sampler): per pixel, every low-discrepancy request is
one scrambled and shuffled (0,2)-sequence over the pixel's samples.

qmc.h (radicalInverse2Single / sobol2Single / sample02Single) is not in the checkout and the reference draws its scrambles and its
shuffle from SFMT, so values cannot be compared with Mitsuba's: parity unpinned.  Tested instead: the defining property of the
point sets (every elementary interval of a scrambled (0,2)-net holds exactly one point), library == oracle bit for bit, images
on the same streams, and that the estimator stays unbiased while its error drops."""
import ctypes as C
import numpy as np
import pytest
import kkdirect
import kkoracle
from kkdirect import Scene, abi, look_at, scenes
import helpers


def _lib_points(seed, pixel, kind, index, array_n, set_size, count):
    lib = kkdirect.load_library()
    out = np.zeros(count * (1 if kind == 1 else 2 * (array_n if kind == 2 else 1)), np.float32)
    assert lib.kkd_probe_ldsampler(seed, pixel, kind, index, array_n, set_size, count, out.ctypes.data_as(abi.pf32)) == 0
    return out if kind == 1 else out.reshape((count, 2) if kind == 0 else (count, array_n, 2))


@pytest.mark.parametrize("kind,index,array_n,set_size", [(0, 0, 1, 16), (0, 3, 1, 64), (1, 0, 1, 32), (1, 2, 1, 4), (2, 0, 4, 16), (2, 1, 3, 8), (0, 1, 1, 1)])
def test_library_and_oracle_hand_out_the_same_points(kind, index, array_n, set_size):
    for seed, pixel in ((1, 0), (0x1234567890ABCDEF, 77777)):
        a = _lib_points(seed, pixel, kind, index, array_n, set_size, set_size)
        b = kkoracle.ldsampler(seed, pixel, kind, index, array_n, set_size, set_size)
        np.testing.assert_array_equal(a, b)
        assert (a >= 0).all() and (a < 1).all()


@pytest.mark.parametrize("n", [4, 16, 64, 256])
def test_two_dimensional_sets_are_scrambled_02_nets(n):
    """all n points of a stream: one per cell of every a x b grid with a * b = n, a and b powers of two (ldsampler.cpp:162-183:
    sample02 with XOR scrambling keeps the net property, the shuffle only reorders)"""
    for pixel in (0, 5, 123456):
        for pts in (_lib_points(9, pixel, 0, 1, 1, n, n), _lib_points(9, pixel, 2, 0, 4, n // 4, n // 4).reshape(-1, 2)):
            a = 1
            while a <= n:
                b = n // a
                cells = (np.floor(pts[:, 0] * a).astype(int) * b + np.floor(pts[:, 1] * b).astype(int))
                assert len(np.unique(cells)) == n, (pixel, a, b)
                a *= 2


def test_one_dimensional_sets_are_stratified_and_streams_differ():
    n = 64
    v = _lib_points(3, 11, 1, 0, 1, n, n)
    assert len(np.unique(np.floor(v * n).astype(int))) == n                       # one point per stratum of width 1/n
    assert not np.array_equal(np.sort(v), v)                                      # shuffled: sample j does not get stratum j
    w = _lib_points(3, 11, 1, 1, 1, n, n)
    assert not np.array_equal(v, w) and abs(np.corrcoef(v, w)[0, 1]) < 0.5        # the next request is another scramble and order
    assert not np.array_equal(v, _lib_points(3, 12, 1, 0, 1, n, n))               # as is the next pixel


def test_sample_split_shares_one_point_set():
    """a rank that renders samples [first, first + spp) of a pixel reads its slice of the pixel's one point set"""
    full = _lib_points(5, 9, 0, 2, 1, 32, 32)
    assert len(np.unique(np.floor(full[8:24, 0] * 32).astype(int))) == 16


def _cornell(w=64, h=64):
    return scenes.cornell_box(w, h)


@pytest.mark.gpu
@pytest.mark.parametrize("integrator,nb,nl,kw", [(abi.INTEGRATOR_KK, 1, 1, {}), (abi.INTEGRATOR_KK, 2, 4, {}), (abi.INTEGRATOR_KK, 3, 1, {}),
                                                 (abi.INTEGRATOR_DIRECT, 1, 1, {}), (abi.INTEGRATOR_DIRECT, 4, 2, {}), (abi.INTEGRATOR_DIRECT, 0, 2, {}),
                                                 (abi.INTEGRATOR_PATH, 1, 1, dict(max_depth=6, rr_depth=2)),
                                                 (abi.INTEGRATOR_PATH, 1, 1, dict(max_depth=-1, rr_depth=3, sampler_dimension=16))])
def test_ldsampler_images_match_the_oracle(gpu, integrator, nb, nl, kw):
    sc = _cornell()
    args = dict(bsdf_samples=nb, light_samples=nl, integrator=integrator, sampler=abi.SAMPLER_LDSAMPLER, **kw)
    want, _ = kkoracle.Oracle(sc).render(spp=16, seed=4, **args)
    with gpu.Context(device=0, seed=4).upload(sc) as ctx:
        got, _ = ctx.render(16, **args)
    assert np.isfinite(got).all()
    assert helpers.rel_mse(helpers.develop(got), helpers.develop(want)) < (7e-4 if kw.get("max_depth") == -1 else 1e-5)


@pytest.mark.gpu
def test_ldsampler_is_unbiased_and_converges_faster(gpu):
    """diffuse floor under a square light, 16 spp: the image mean agrees with the independent sampler's 1024-spp mean, and
    the per-pixel error against that reference is clearly below the independent sampler's at equal sample count"""
    sc = _cornell(48, 48)
    with gpu.Context(device=0, seed=7).upload(sc) as ctx:
        ref = helpers.develop(ctx.render(1024)[0])
        ind = helpers.develop(ctx.render(16, sample_first=4096)[0])
        ld = helpers.develop(ctx.render(16, sampler=abi.SAMPLER_LDSAMPLER)[0])
    err_ind, err_ld = np.sqrt(((ind - ref) ** 2).mean()), np.sqrt(((ld - ref) ** 2).mean())
    assert abs(ld.mean() - ref.mean()) < 0.01 * ref.mean()
    assert err_ld < 0.8 * err_ind, (err_ld, err_ind)


@pytest.mark.gpu
def test_sample_split_renders_the_same_image(gpu):
    """two calls with 8 samples each of a 16-sample ldsampler render (sample_count_total = 16) add up to the one-call image"""
    sc = _cornell(32, 32)
    with gpu.Context(device=0, seed=3).upload(sc) as ctx:
        whole, _ = ctx.render(16, sampler=abi.SAMPLER_LDSAMPLER)
        a, _ = ctx.render(8, sampler=abi.SAMPLER_LDSAMPLER, sample_first=0, sample_count_total=16)
        b, _ = ctx.render(8, sampler=abi.SAMPLER_LDSAMPLER, sample_first=8, sample_count_total=16)
    np.testing.assert_allclose(a + b, whole, rtol=1e-5, atol=1e-6)


def test_host_mirror_of_the_sampler_plugins():
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        s = kkdirect.LowDiscrepancySampler(sampleCount=12)            # ldsampler.cpp:83-87
    assert s.sampleCount == 16 and len(w) == 1 and "power of two" in str(w[0].message)
    assert kkdirect.LowDiscrepancySampler().render_args() == dict(sampler=abi.SAMPLER_LDSAMPLER, sampler_dimension=4)
    assert kkdirect.IndependentSampler(sampleCount=7).sampleCount == 7
    with pytest.raises(ValueError):
        kkdirect.LowDiscrepancySampler(sampleCount=4, foo=1)


@pytest.mark.gpu
def test_render_params_are_validated(gpu):
    sc = _cornell(16, 16)
    with gpu.Context(device=0, seed=1).upload(sc) as ctx:
        with pytest.raises(Exception, match="unknown sampler"):
            ctx.render(4, sampler=kkdirect.abi.SAMPLER_SOBOL + 1)
        with pytest.raises(Exception, match="sample_count_total"):
            ctx.render(8, sample_first=12, sample_count_total=16)
        film, _ = kkdirect.DirectIntegrator().render(sc, ctx=ctx, sampler=kkdirect.LowDiscrepancySampler(sampleCount=8))
        assert np.isfinite(film).all() and film[..., 4].sum() > 0
