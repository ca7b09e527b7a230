"""sobol sampler (reference src/samplers/sobol.cpp:82-279, src/samplers/sobolseq.h:43-60, 102-137, direction numbers of
src/samplers/sobolseq.cpp = S. Joe and F. Y. Kuo, new-joe-kuo-6.21201).

Deterministic (scramble = 0), so the values ARE Mitsuba's, and here they are pinned by the reference's own data: the library
regenerates the 1024 x 52 generator matrices from the published direction numbers with its own code and must reproduce the
CRC-32 of the reference's table; `sampleSingle` and `look_up` are held bit for bit against known answers that
scripts/make_sobol_dirnums.py computed on the reference's tables (tests/golden/sobol_kat.json).  Then the defining properties
(every sample of a pixel lands inside it; the first two dimensions of 2^k consecutive points are a (0, k, 2)-net), the layout of the
sample arrays, library == oracle (whose look_up is a different piece of code), the host mirror, and images on the GPU."""
import ctypes as C
import json
import os
import numpy as np
import pytest
import kkdirect
import kkoracle
from kkdirect import abi, scenes
import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "sobol_kat.json")))


def _raw(index=0, dim=0, m=0, frame=0, px=0, py=0):
    bits, idx, crc = abi.u32(), abi.u64(), abi.u32()
    rc = kkdirect.load_library().kkd_probe_sobol_raw(index, dim, m, frame, px, py, C.byref(bits), C.byref(idx) if m else None, C.byref(crc))
    assert rc == 0, rc
    return bits.value, idx.value, crc.value


def _lib(res, xy, kind, dim, array_n=1, n_arrays=0, count=8):
    out = np.zeros(count * (1 if kind == 1 else 2 * (array_n if kind == 2 else 1)), np.float32)
    assert kkdirect.load_library().kkd_probe_sobol(res[0], res[1], xy[0], xy[1], kind, dim, array_n, n_arrays, count, out.ctypes.data_as(abi.pf32)) == 0
    return out if kind == 1 else out.reshape((count, 2) if kind == 0 else (count, array_n, 2))


def test_regenerated_matrices_have_the_reference_tables_checksum():
    assert _raw()[2] == KAT["matrices32_crc32"]


def test_sample_single_known_answers():
    for index, dim, want in KAT["sample"]:
        assert _raw(index, dim)[0] == want, (index, dim)
        assert kkoracle.sobol_bits(index, dim) == want, (index, dim)
    for i, row in enumerate(KAT["first_points"]):
        assert [_raw(i, d)[0] for d in range(8)] == row
    # the published beginning of the sequence: dimension 0 is van der Corput, dimension 1 the first Sobol' dimension
    assert [_raw(i, 0)[0] / 2.0 ** 32 for i in range(5)] == [0.0, 0.5, 0.25, 0.75, 0.125]
    assert [_raw(i, 1)[0] / 2.0 ** 32 for i in range(5)] == [0.0, 0.5, 0.75, 0.25, 0.625]       # m = 1, 3, 5: columns 1/2, 3/4, 5/8


def test_look_up_known_answers():
    for m, frame, px, py, want in KAT["look_up"]:
        assert _raw(m=m, frame=frame, px=px, py=py)[1] == want, (m, frame, px, py)
        assert kkoracle.sobol_look_up(m, frame, px, py) == want, (m, frame, px, py)


@pytest.mark.parametrize("res", [(200, 150), (64, 48), (1920, 1080), (5, 3), (3840, 2160)])
def test_every_sample_of_a_pixel_lands_inside_it(res):
    """sobol.cpp:216-228, 254-256: sample j of pixel (x, y) is the j-th point of the sequence whose first two coordinates, scaled by
    the power-of-two resolution, fall into that pixel"""
    rng = np.random.RandomState(res[0])
    for _ in range(12):
        x, y = int(rng.randint(res[0])), int(rng.randint(res[1]))
        p = _lib(res, (x, y), 0, 0, count=32)
        assert (p >= 0).all() and (p < 1).all(), (x, y, p.min(), p.max())
    # consecutive samples of one pixel refine it like a (0, 2)-sequence: 16 samples occupy every cell of every 16-cell dyadic partition
    p = _lib(res, (res[0] // 2, res[1] // 3), 0, 0, count=16)
    for a, b in ((16, 1), (8, 2), (4, 4), (2, 8), (1, 16)):
        cells = (np.floor(p[:, 0] * a) * b + np.floor(p[:, 1] * b)).astype(int)
        assert sorted(cells) == list(range(16)), (a, b)


def test_pixels_own_disjoint_indices_and_tiny_films_use_the_plain_sequence():
    seen = set()
    for x in range(8):
        for y in range(8):
            for frame in range(4):
                idx = _raw(m=3, frame=frame, px=x, py=y)[1]
                assert idx >> 6 == frame and idx not in seen
                seen.add(idx)
    assert len(seen) == 8 * 8 * 4
    # a 2 x 2 film has m = 1: sobol.cpp:221 only looks pixels up when m > 1, the samples are points 0, 1, 2, ... themselves
    p = _lib((2, 2), (1, 1), 0, 0, count=4)
    np.testing.assert_array_equal(p, np.array([[0, 0], [0.5, 0.5], [0.25, 0.75], [0.75, 0.25]], np.float32))


def test_sample_arrays_and_the_skip_over_their_dimensions():
    n = 3
    res, xy = (64, 48), (17, 9)
    arr = _lib(res, xy, 2, 0, array_n=n, n_arrays=2, count=5)        # array 0: dimensions 5, 6 of point look_up(j), j = s * n + e
    for s in range(5):
        for e in range(n):
            idx = _raw(m=6, frame=s * n + e, px=xy[0], py=xy[1])[1]
            want = [min(_raw(idx, d)[0] * 2.0 ** -32, 1 - 2.0 ** -24) for d in (5, 6)]
            np.testing.assert_allclose(arr[s, e], want, rtol=0, atol=1e-7)
    arr1 = _lib(res, xy, 2, 1, array_n=n, n_arrays=2, count=5)       # array 1: dimensions 7, 8
    idx = _raw(m=6, frame=5, px=xy[0], py=xy[1])[1]
    np.testing.assert_allclose(arr1[1, 2], [_raw(idx, 7)[0] * 2.0 ** -32, _raw(idx, 8)[0] * 2.0 ** -32], atol=1e-7)
    # next2D at a running dimension inside [5, 5 + 2 * arrays) moves to the end of the reserved range (sobol.cpp:246-247) ...
    assert np.array_equal(_lib(res, xy, 0, 6, n_arrays=2), _lib(res, xy, 0, 9, n_arrays=0))
    # ... but one that starts at dimension 4 takes 4 and 5 although 5 is reserved: the check precedes the request
    assert np.array_equal(_lib(res, xy, 0, 4, n_arrays=2), _lib(res, xy, 0, 4, n_arrays=0))


@pytest.mark.parametrize("args", [((640, 480), (3, 5), 0, 0), ((640, 480), (300, 205), 0, 2), ((64, 48), (63, 47), 1, 7),
                                  ((1920, 1080), (1000, 700), 0, 500), ((3840, 2160), (3839, 2159), 0, 1022), ((640, 480), (-2, -1), 0, 0), ((640, 480), (5, -1), 0, 0)])
def test_library_and_oracle_hand_out_the_same_points(args):
    res, xy, kind, dim = args
    a = _lib(res, xy, kind, dim, count=64)
    b = kkoracle.sobol(res[0], res[1], xy[0], xy[1], kind, dim, 1, 0, 64)
    np.testing.assert_array_equal(a, b)
    assert (a >= 0).all() and (a < 1).all()
    arr = _lib(res, xy, 2, 1, array_n=4, n_arrays=2, count=16)
    np.testing.assert_array_equal(arr, kkoracle.sobol(res[0], res[1], xy[0], xy[1], 2, 1, 4, 2, 16))


def test_probe_rejects_bad_arguments_and_host_mirror():
    lib = kkdirect.load_library(); out = np.zeros(8, np.float32); p = out.ctypes.data_as(abi.pf32)
    assert lib.kkd_probe_sobol(0, 4, 0, 0, 0, 0, 1, 0, 1, p) != 0             # empty film
    assert lib.kkd_probe_sobol(4, 4, 0, 0, 0, 1023, 1, 0, 1, p) != 0          # next2D needs two dimensions of the table
    assert lib.kkd_probe_sobol(4, 4, 0, 0, 1, 1024, 1, 0, 1, p) != 0
    assert lib.kkd_probe_sobol(4, 4, 0, 0, 2, 1, 2, 1, 1, p) != 0             # array 1 of one requested array
    assert kkdirect.SobolSampler().render_args() == dict(sampler=abi.SAMPLER_SOBOL)
    assert kkdirect.SobolSampler(sampleCount=7).sampleCount == 7
    with pytest.raises(ValueError):
        kkdirect.SobolSampler(scramble=5)                          # sampleTEA-scrambled sequences are not offered
    with pytest.raises(ValueError):
        kkdirect.SobolSampler(foo=1)


@pytest.mark.gpu
@pytest.mark.parametrize("integrator,nb,nl,kw", [(abi.INTEGRATOR_KK, 1, 1, {}), (abi.INTEGRATOR_KK, 2, 4, {}), (abi.INTEGRATOR_KK, 3, 1, {}),
                                                 (abi.INTEGRATOR_DIRECT, 1, 1, {}), (abi.INTEGRATOR_DIRECT, 4, 2, {}), (abi.INTEGRATOR_DIRECT, 0, 2, {}),
                                                 (abi.INTEGRATOR_PATH, 1, 1, dict(max_depth=6, rr_depth=2)),
                                                 (abi.INTEGRATOR_PATH, 1, 1, dict(max_depth=-1, rr_depth=3))])
def test_sobol_images_match_the_oracle(gpu, integrator, nb, nl, kw):
    sc = scenes.cornell_box(64, 64)
    args = dict(bsdf_samples=nb, light_samples=nl, integrator=integrator, sampler=abi.SAMPLER_SOBOL, **kw)
    want, _ = kkoracle.Oracle(sc).render(spp=16, seed=4, **args)
    with gpu.Context(device=0, seed=4).upload(sc) as ctx:
        got, _ = ctx.render(16, **args)
    assert np.isfinite(got).all()
    assert helpers.rel_mse(helpers.develop(got), helpers.develop(want)) < (7e-4 if kw.get("max_depth") == -1 else 1e-5)


@pytest.mark.gpu
def test_sobol_is_deterministic_unbiased_converges_faster_and_splits(gpu):
    sc = scenes.cornell_box(48, 48)
    with gpu.Context(device=0, seed=3).upload(sc) as ctx:
        ref = helpers.develop(ctx.render(1024)[0])
        ind = helpers.develop(ctx.render(16, sample_first=4096)[0])
        sob = helpers.develop(ctx.render(16, sampler=abi.SAMPLER_SOBOL)[0])
        whole, _ = ctx.render(16, sampler=abi.SAMPLER_SOBOL)
        a, _ = ctx.render(8, sampler=abi.SAMPLER_SOBOL, sample_first=0, sample_count_total=16)
        b, _ = ctx.render(8, sampler=abi.SAMPLER_SOBOL, sample_first=8, sample_count_total=16)
        film, _ = kkdirect.DirectIntegrator().render(sc, ctx=ctx, sampler=kkdirect.SobolSampler(sampleCount=5))
    with gpu.Context(device=0, seed=99).upload(sc) as ctx:                      # the seed plays no part
        other = helpers.develop(ctx.render(16, sampler=abi.SAMPLER_SOBOL)[0])
    np.testing.assert_allclose(other, sob, rtol=1e-5, atol=1e-7)
    err_ind, err_sob = np.sqrt(((ind - ref) ** 2).mean()), np.sqrt(((sob - ref) ** 2).mean())
    assert abs(sob.mean() - ref.mean()) < 0.02 * ref.mean()
    assert err_sob < 0.9 * err_ind, (err_sob, err_ind)
    np.testing.assert_allclose(a + b, whole, rtol=1e-5, atol=1e-6)
    assert np.isfinite(film).all() and film[..., 4].sum() > 0
    # high-quality edges: samples left of / above the film take the plain sequence (sobol.cpp:221) and still splat into it
    sc.sensor.film_high_quality_edges = 1
    want, _ = kkoracle.Oracle(sc).render(spp=8, seed=1, sampler=abi.SAMPLER_SOBOL)
    with gpu.Context(device=0, seed=1).upload(sc) as ctx:
        got, _ = ctx.render(8, sampler=abi.SAMPLER_SOBOL)
    assert helpers.rel_mse(helpers.develop(got), helpers.develop(want)) < 1e-5
