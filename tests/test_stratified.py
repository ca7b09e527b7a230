"""stratified sampler (reference src/samplers/stratified.cpp:147-207): per pixel, every stratified 2D (1D) request visits each cell of
a res x res grid (each of res^2 intervals) once over the pixel's res^2 samples, in a shuffled order, jittered inside the cell;
requested arrays are Latin hypercubes over all res^2 * n entries (libcore/util.cpp:565-576).

The reference fills its permutation tables and draws its jitter from SFMT, so values cannot be compared with Mitsuba's: parity
unpinned.  Tested instead: the defining properties of the point sets, library == oracle bit for bit, images on the same
streams, and that the estimator stays unbiased while its error drops."""
import numpy as np
import pytest
import kkdirect
import kkoracle
from kkdirect import abi, scenes
import helpers


def _lib_points(seed, pixel, kind, index, array_n, res, flat_dim, count):
    lib = kkdirect.load_library()
    out = np.zeros(count * (1 if kind == 1 else 2 * (array_n if kind == 2 else 1)), np.float32)
    assert lib.kkd_probe_stratified(seed, pixel, kind, index, array_n, res, flat_dim, count, out.ctypes.data_as(abi.pf32)) == 0
    return out if kind == 1 else out.reshape((count, 2) if kind == 0 else (count, array_n, 2))


@pytest.mark.parametrize("kind,index,array_n,res,flat", [(0, 0, 1, 4, 0), (0, 3, 1, 7, 6), (1, 0, 1, 5, 4), (1, 2, 1, 2, 10), (2, 0, 4, 3, 2), (2, 1, 3, 4, 8),
                                                         (0, 1, 1, 1, 2), (0, 0, 1, 3, 0)])
def test_library_and_oracle_hand_out_the_same_points(kind, index, array_n, res, flat):
    for seed, pixel in ((1, 0), (0x1234567890ABCDEF, 77777)):
        a = _lib_points(seed, pixel, kind, index, array_n, res, flat, res * res)
        b = kkoracle.stratified(seed, pixel, kind, index, array_n, res, flat, res * res)
        np.testing.assert_array_equal(a, b)
        assert (a >= 0).all() and (a < 1).all()


@pytest.mark.parametrize("res", [1, 2, 3, 8, 13])
def test_two_dimensional_requests_fill_every_cell_once(res):
    """stratified.cpp:191-200: cell k of the grid, k a permutation of the pixel's sample indices"""
    for pixel in (0, 5, 123456):
        pts = _lib_points(9, pixel, 0, 1, 1, res, 2, res * res)
        cells = np.floor(pts[:, 0] * res).astype(int) + res * np.floor(pts[:, 1] * res).astype(int)
        assert sorted(cells) == list(range(res * res))
        if res >= 8:
            assert not np.array_equal(cells, np.arange(res * res))               # shuffled
            frac = pts * res - np.floor(pts * res)                               # the offsets inside the cells are uniform
            assert abs(frac.mean() - 0.5) < 0.1 and frac.std() > 0.2


def test_one_dimensional_requests_fill_every_interval_once_and_streams_differ():
    res = 8; n = res * res
    v = _lib_points(3, 11, 1, 0, 1, res, 4, n)                                     # stratified.cpp:181-186
    assert sorted(np.floor(v * n).astype(int)) == list(range(n))
    assert not np.array_equal(np.sort(v), v)
    w = _lib_points(3, 11, 1, 1, 1, res, 4, n)
    assert not np.array_equal(np.floor(v * n), np.floor(w * n))                   # the next request is shuffled differently
    assert not np.array_equal(np.floor(v * n), np.floor(_lib_points(3, 12, 1, 0, 1, res, 4, n) * n))   # as is the next pixel


@pytest.mark.parametrize("res,array_n", [(2, 4), (4, 3), (5, 2)])
def test_arrays_are_latin_hypercubes(res, array_n):
    """libcore/util.cpp:565-576: over all res^2 * n entries every coordinate visits each of the res^2 * n intervals once, and the two
    coordinates are shuffled independently"""
    n = res * res * array_n
    pts = _lib_points(21, 4, 2, 0, array_n, res, 2, res * res).reshape(-1, 2)
    ix, iy = np.floor(pts[:, 0] * n).astype(int), np.floor(pts[:, 1] * n).astype(int)
    assert sorted(ix) == list(range(n)) and sorted(iy) == list(range(n))
    assert not np.array_equal(ix, iy)
    other = _lib_points(21, 4, 2, 1, array_n, res, 2, res * res).reshape(-1, 2)
    assert not np.array_equal(np.floor(other[:, 0] * n).astype(int), ix)          # the second requested array is its own hypercube


def test_values_stay_below_one():
    """(k + xi) * (1 / res) rounds to 1 in single precision for some res (3, with xi = 1 - 2^-23); capped, see kkd_math.h"""
    for res in (3, 5, 6, 7):
        v = _lib_points(77, 1, 0, 0, 1, res, 0, res * res)
        assert v.max() < 1.0
    assert np.float32(np.float32(2) + np.float32(1 - 2.0 ** -23)) * np.float32(1.0 / 3.0) == np.float32(1.0)      # what the cap is for


def test_probe_rejects_bad_arguments():
    lib = kkdirect.load_library(); out = np.zeros(8, np.float32); p = out.ctypes.data_as(abi.pf32)
    assert lib.kkd_probe_stratified(1, 0, 0, 0, 1, 0, 0, 1, p) != 0               # no cells
    assert lib.kkd_probe_stratified(1, 0, 0, 0, 1, 2, 1, 1, p) != 0               # odd dimension of the jitter
    assert lib.kkd_probe_stratified(1, 0, 2, 0, 0, 2, 0, 1, p) != 0               # empty array
    assert lib.kkd_probe_stratified(1, 0, 3, 0, 1, 2, 0, 1, p) != 0               # unknown kind


def _cornell(w=64, h=64):
    return scenes.cornell_box(w, h)


@pytest.mark.gpu
@pytest.mark.parametrize("integrator,nb,nl,kw", [(abi.INTEGRATOR_KK, 1, 1, {}), (abi.INTEGRATOR_KK, 2, 4, {}), (abi.INTEGRATOR_KK, 3, 1, {}),
                                                 (abi.INTEGRATOR_DIRECT, 1, 1, {}), (abi.INTEGRATOR_DIRECT, 4, 2, {}), (abi.INTEGRATOR_DIRECT, 0, 2, {}),
                                                 (abi.INTEGRATOR_PATH, 1, 1, dict(max_depth=6, rr_depth=2)),
                                                 (abi.INTEGRATOR_PATH, 1, 1, dict(max_depth=-1, rr_depth=3, sampler_dimension=16))])
def test_stratified_images_match_the_oracle(gpu, integrator, nb, nl, kw):
    sc = _cornell()
    args = dict(bsdf_samples=nb, light_samples=nl, integrator=integrator, sampler=abi.SAMPLER_STRATIFIED, **kw)
    want, _ = kkoracle.Oracle(sc).render(spp=16, seed=4, **args)
    with gpu.Context(device=0, seed=4).upload(sc) as ctx:
        got, _ = ctx.render(16, **args)
    assert np.isfinite(got).all()
    assert helpers.rel_mse(helpers.develop(got), helpers.develop(want)) < (7e-4 if kw.get("max_depth") == -1 else 1e-5)


@pytest.mark.gpu
def test_stratified_is_unbiased_and_converges_faster(gpu):
    """16 spp against the independent sampler's 1024-spp image.  The image mean of one 16-spp render scatters by about 1 % (oracle,
    12 seeds: -2.3 % ... +1.1 %, mean -0.3 %; the independent sampler's: -3.7 % ... +7.5 %), hence the seed and the 1.5 % bound."""
    sc = _cornell(48, 48)
    with gpu.Context(device=0, seed=3).upload(sc) as ctx:
        ref = helpers.develop(ctx.render(1024)[0])
        ind = helpers.develop(ctx.render(16, sample_first=4096)[0])
        st = helpers.develop(ctx.render(16, sampler=abi.SAMPLER_STRATIFIED)[0])
    err_ind, err_st = np.sqrt(((ind - ref) ** 2).mean()), np.sqrt(((st - ref) ** 2).mean())
    assert abs(st.mean() - ref.mean()) < 0.015 * ref.mean()
    assert err_st < 0.9 * err_ind, (err_st, err_ind)


@pytest.mark.gpu
def test_sample_split_and_non_square_counts(gpu):
    """two calls with 8 samples each of a 16-sample render add up to the one-call image; 12 samples use the first 12 of the 16
    strata (the Python mirror and the Mitsuba adapter round to 16 first, stratified.cpp:62-72)"""
    sc = _cornell(32, 32)
    with gpu.Context(device=0, seed=3).upload(sc) as ctx:
        whole, _ = ctx.render(16, sampler=abi.SAMPLER_STRATIFIED)
        a, _ = ctx.render(8, sampler=abi.SAMPLER_STRATIFIED, sample_first=0, sample_count_total=16)
        b, _ = ctx.render(8, sampler=abi.SAMPLER_STRATIFIED, sample_first=8, sample_count_total=16)
        c, _ = ctx.render(12, sampler=abi.SAMPLER_STRATIFIED)
        film, _ = kkdirect.DirectIntegrator().render(sc, ctx=ctx, sampler=kkdirect.StratifiedSampler(sampleCount=9))
    np.testing.assert_allclose(a + b, whole, rtol=1e-5, atol=1e-6)
    assert np.isfinite(c).all() and c[..., 4].sum() > 0
    want, _ = kkoracle.Oracle(sc).render(spp=12, seed=3, sampler=abi.SAMPLER_STRATIFIED)
    assert helpers.rel_mse(helpers.develop(c), helpers.develop(want)) < 1e-5
    assert np.isfinite(film).all() and film[..., 4].sum() > 0


def test_host_mirror_of_the_sampler_plugin():
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        s = kkdirect.StratifiedSampler(sampleCount=12)                 # stratified.cpp:62-72
    assert (s.sampleCount, s.resolution) == (16, 4) and len(w) == 1 and "perfect square" in str(w[0].message)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert kkdirect.StratifiedSampler(sampleCount=25).sampleCount == 25 and not w
    assert kkdirect.StratifiedSampler().render_args() == dict(sampler=abi.SAMPLER_STRATIFIED, sampler_dimension=4)
    assert kkdirect.StratifiedSampler().sampleCount == 4
    with pytest.raises(ValueError):
        kkdirect.StratifiedSampler(sampleCount=4, foo=1)
