"""halton sampler (reference src/samplers/halton.cpp:240-384, src/libcore/qmc.cpp:136-150, src/samplers/faure.cpp:101-131).

Unlike the other samplers this one is deterministic (no random stream with scramble = -1 or 0), so its values are those Mitsuba
computes.  No fixture in the reference holds any, so parity stays unpinned by the reference's own data; pinned here by known
answers that do not come from the oracle: radical inverses as exact fractions, Faure's published permutations, the defining
property of the pixel addressing (the first two dimensions of every sample of a pixel fall inside that pixel), and the layout of
the sample arrays.  Then library == oracle bit for bit, and images on the GPU."""
from fractions import Fraction
import ctypes as C
import numpy as np
import pytest
import kkdirect
import kkoracle
from kkdirect import abi, scenes
import helpers

PRIMES = [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43]


def _lib(res, xy, faure, kind, dim, array_n=1, n_arrays=0, count=8):
    lib = kkdirect.load_library()
    out = np.zeros(count * (1 if kind == 1 else 2 * (array_n if kind == 2 else 1)), np.float32)
    assert lib.kkd_probe_halton(res[0], res[1], xy[0], xy[1], int(faure), kind, dim, array_n, n_arrays, count, out.ctypes.data_as(abi.pf32)) == 0
    return out if kind == 1 else out.reshape((count, 2) if kind == 0 else (count, array_n, 2))


def _faure(base):
    out = (C.c_uint16 * base)()
    assert kkdirect.load_library().kkd_probe_faure_permutation(base, out) == 0
    return list(out)


def _radical_inverse(base, i, perm=None):
    """exact: digits of i mirrored about the radix point (each passed through perm)"""
    v, f = Fraction(0), Fraction(1, base)
    while i:
        i, d = divmod(i, base)
        v += (perm[d] if perm else d) * f
        f /= base
    return v


def test_faure_permutations_are_the_published_ones():
    """Faure 1992, "Good permutations for extreme discrepancy": the bases that are prime (the only ones the sampler keeps)"""
    assert _faure(2) == [0, 1] and _faure(3) == [0, 1, 2]
    assert _faure(5) == [0, 3, 2, 1, 4]
    assert _faure(7) == [0, 2, 5, 3, 1, 4, 6]
    for b in (11, 13, 8161):                                   # every one is a permutation that fixes 0 and b - 1
        p = _faure(b)
        assert sorted(p) == list(range(b)) and p[0] == 0 and p[-1] == b - 1
    assert kkdirect.load_library().kkd_probe_faure_permutation(9, (C.c_uint16 * 9)()) != 0      # not a prime: not kept


@pytest.mark.parametrize("faure", [False, True])
def test_radical_inverses_are_exact_fractions(faure):
    """a 1 x 1 film has stride 1 and offset 0 (halton.cpp:252-260): sample j is index j of the sequence"""
    n = 200
    for d, base in enumerate(PRIMES):
        got = _lib((1, 1), (0, 0), faure, 1, d, count=n)
        perm = _faure(base) if faure else None
        want = np.array([float(_radical_inverse(base, j, perm)) for j in range(n)])
        np.testing.assert_allclose(got, want, rtol=3e-7, atol=0)
    assert _lib((1, 1), (0, 0), False, 1, 0, count=4).tolist() == [0.0, 0.5, 0.25, 0.75]       # van der Corput
    v = _lib((1, 1), (0, 0), False, 0, 2, count=4)                                             # next2D at dimension 2: bases 5 and 7
    assert np.allclose(v, [[0, 0], [1 / 5, 1 / 7], [2 / 5, 2 / 7], [3 / 5, 3 / 7]], rtol=3e-7)


@pytest.mark.parametrize("res", [(200, 150), (64, 48), (1920, 1080), (5, 3)])
def test_every_sample_of_a_pixel_lands_inside_it(res):
    """halton.cpp:282-294, 375-377: the offset found through the Chinese remainder theorem makes dimensions 0 and 1, scaled to the
    tile of pow2 x pow3 pixels, fall into the pixel; the fractions left over are themselves well spread"""
    rng = np.random.RandomState(3)
    for _ in range(40):
        x, y = int(rng.randint(res[0])), int(rng.randint(res[1]))
        for faure in (False, True):
            v = _lib(res, (x, y), faure, 0, 0, count=16)
            assert (v >= 0).all() and (v < 1).all(), (x, y, v)
    v = _lib(res, (res[0] // 2, res[1] // 2), True, 0, 0, count=64)
    assert len(np.unique(np.floor(v[:, 0] * 64))) == 64            # base 2 in the leftover digits: one sample per 1/64 column
    far = _lib(res, (res[0] // 2 + 128, res[1] // 2), True, 0, 0, count=4)
    assert np.array_equal(far, v[:4])                              # the sequence is tiled with a period of 128 pixels (MAX_RESOLUTION)


def test_pixels_of_a_tile_use_disjoint_indices():
    """distinct pixels of one tile own distinct residues modulo the stride: their positions in the (2, 3) plane never coincide"""
    res = (16, 9)                                                  # tile 16 x 9, stride 144
    seen = set()
    for y in range(9):
        for x in range(16):
            v = _lib(res, (x, y), False, 0, 0, count=2)
            for j in range(2):
                key = (round((x + float(v[j, 0])) / 16 * 2 ** 20), round((y + float(v[j, 1])) / 9 * 3 ** 12))
                assert key not in seen
                seen.add(key)


def test_sample_arrays_and_the_skip_over_their_dimensions():
    n = 3
    arr = _lib((1, 1), (0, 0), False, 2, 0, array_n=n, n_arrays=2, count=5)        # array 0: dimensions 5, 6 = bases 13, 17; entry j = s * n + e
    for s in range(5):
        for e in range(n):
            j = s * n + e
            assert np.allclose(arr[s, e], [float(_radical_inverse(13, j)), float(_radical_inverse(17, j))], rtol=3e-7)
    arr1 = _lib((1, 1), (0, 0), False, 2, 1, array_n=n, n_arrays=2, count=5)       # array 1: dimensions 7, 8 = bases 19, 23
    assert np.allclose(arr1[1, 2], [float(_radical_inverse(19, 5)), float(_radical_inverse(23, 5))], rtol=3e-7)
    # next2D at a running dimension inside [5, 5 + 2 * arrays) moves to the end of the reserved range (halton.cpp:366-367) ...
    assert np.array_equal(_lib((1, 1), (0, 0), True, 0, 6, n_arrays=2), _lib((1, 1), (0, 0), True, 0, 9, n_arrays=0))
    # ... but one that starts at dimension 4 takes 4 and 5 although 5 is reserved: the check precedes the request
    assert np.array_equal(_lib((1, 1), (0, 0), True, 0, 4, n_arrays=2), _lib((1, 1), (0, 0), True, 0, 4, n_arrays=0))


@pytest.mark.parametrize("args", [((640, 480), (3, 5), False, 0, 0), ((640, 480), (300, 205), True, 0, 2), ((64, 48), (63, 47), True, 1, 7),
                                  ((1920, 1080), (1000, 700), True, 0, 500), ((3840, 2160), (3839, 2159), True, 0, 1022), ((640, 480), (-2, -1), True, 0, 0)])
def test_library_and_oracle_hand_out_the_same_points(args):
    res, xy, faure, kind, dim = args
    a = _lib(res, xy, faure, kind, dim, count=64)
    b = kkoracle.halton(res[0], res[1], xy[0], xy[1], faure, kind, dim, 1, 0, 64)
    np.testing.assert_array_equal(a, b)
    assert (a >= 0).all() and (a < 1).all()
    arr = _lib(res, xy, faure, 2, 1, array_n=4, n_arrays=2, count=16)
    np.testing.assert_array_equal(arr, kkoracle.halton(res[0], res[1], xy[0], xy[1], faure, 2, 1, 4, 2, 16))


def test_probe_rejects_bad_arguments():
    lib = kkdirect.load_library(); out = np.zeros(8, np.float32); p = out.ctypes.data_as(abi.pf32)
    assert lib.kkd_probe_halton(0, 4, 0, 0, 1, 0, 0, 1, 0, 1, p) != 0            # empty film
    assert lib.kkd_probe_halton(4, 4, 0, 0, 1, 0, 1023, 1, 0, 1, p) != 0         # next2D needs two dimensions of the prime table
    assert lib.kkd_probe_halton(4, 4, 0, 0, 1, 1, 1024, 1, 0, 1, p) != 0
    assert lib.kkd_probe_halton(4, 4, 0, 0, 1, 2, 1, 2, 1, 1, p) != 0            # array 1 of one requested array
    assert lib.kkd_probe_halton(4, 4, 0, 0, 1, 3, 0, 1, 0, 1, p) != 0


def test_host_mirror_of_the_sampler_plugin():
    assert kkdirect.HaltonSampler().render_args() == dict(sampler=abi.SAMPLER_HALTON)          # scramble = -1 by default, halton.cpp:125
    assert kkdirect.HaltonSampler(sampleCount=7, scramble=0).render_args() == dict(sampler=abi.SAMPLER_HALTON_PLAIN)
    assert kkdirect.HaltonSampler(sampleCount=7).sampleCount == 7
    with pytest.raises(ValueError):
        kkdirect.HaltonSampler(scramble=5)                         # random permutations come from Mitsuba's SFMT
    with pytest.raises(ValueError):
        kkdirect.HaltonSampler(foo=1)


def _cornell(w=64, h=64):
    return scenes.cornell_box(w, h)


@pytest.mark.gpu
@pytest.mark.parametrize("sampler", [abi.SAMPLER_HALTON, abi.SAMPLER_HALTON_PLAIN])
@pytest.mark.parametrize("integrator,nb,nl,kw", [(abi.INTEGRATOR_KK, 1, 1, {}), (abi.INTEGRATOR_KK, 2, 4, {}), (abi.INTEGRATOR_KK, 3, 1, {}),
                                                 (abi.INTEGRATOR_DIRECT, 1, 1, {}), (abi.INTEGRATOR_DIRECT, 4, 2, {}), (abi.INTEGRATOR_DIRECT, 0, 2, {}),
                                                 (abi.INTEGRATOR_PATH, 1, 1, dict(max_depth=6, rr_depth=2)),
                                                 (abi.INTEGRATOR_PATH, 1, 1, dict(max_depth=-1, rr_depth=3))])
def test_halton_images_match_the_oracle(gpu, sampler, integrator, nb, nl, kw):
    sc = _cornell()
    args = dict(bsdf_samples=nb, light_samples=nl, integrator=integrator, sampler=sampler, **kw)
    want, _ = kkoracle.Oracle(sc).render(spp=16, seed=4, **args)
    with gpu.Context(device=0, seed=4).upload(sc) as ctx:
        got, _ = ctx.render(16, **args)
    assert np.isfinite(got).all()
    assert helpers.rel_mse(helpers.develop(got), helpers.develop(want)) < (7e-4 if kw.get("max_depth") == -1 else 1e-5)


@pytest.mark.gpu
def test_halton_is_deterministic_unbiased_and_converges_faster(gpu):
    sc = _cornell(48, 48)
    with gpu.Context(device=0, seed=3).upload(sc) as ctx:
        ref = helpers.develop(ctx.render(1024)[0])
        ind = helpers.develop(ctx.render(16, sample_first=4096)[0])
        hal = helpers.develop(ctx.render(16, sampler=abi.SAMPLER_HALTON)[0])
    with gpu.Context(device=0, seed=99).upload(sc) as ctx:                      # the seed plays no part
        other = helpers.develop(ctx.render(16, sampler=abi.SAMPLER_HALTON)[0])
    np.testing.assert_allclose(other, hal, rtol=1e-5, atol=1e-7)
    err_ind, err_hal = np.sqrt(((ind - ref) ** 2).mean()), np.sqrt(((hal - ref) ** 2).mean())
    assert abs(hal.mean() - ref.mean()) < 0.02 * ref.mean()
    assert err_hal < 0.9 * err_ind, (err_hal, err_ind)


@pytest.mark.gpu
def test_sample_split_and_edges(gpu):
    sc = _cornell(32, 32)
    with gpu.Context(device=0, seed=3).upload(sc) as ctx:
        whole, _ = ctx.render(16, sampler=abi.SAMPLER_HALTON)
        a, _ = ctx.render(8, sampler=abi.SAMPLER_HALTON, sample_first=0, sample_count_total=16)
        b, _ = ctx.render(8, sampler=abi.SAMPLER_HALTON, sample_first=8, sample_count_total=16)
        film, _ = kkdirect.DirectIntegrator().render(sc, ctx=ctx, sampler=kkdirect.HaltonSampler(sampleCount=5))
        with pytest.raises(Exception, match="unknown sampler"):
            ctx.render(4, sampler=9)
    np.testing.assert_allclose(a + b, whole, rtol=1e-5, atol=1e-6)
    assert np.isfinite(film).all() and film[..., 4].sum() > 0


# ---- hammersley (reference src/samplers/hammersley.cpp:181-283): the same radical inverses, dimension 0 = index / (samples of a tile) ----

HAM = 0xFFFFFFFF          # kkd_probe_halton: n_arrays value that selects the hammersley plugin, array_n = its sampleCount


@pytest.mark.parametrize("res,count", [((200, 150), 16), ((64, 48), 4), ((1920, 1080), 64), ((5, 3), 7)])
def test_hammersley_samples_land_inside_their_pixel(res, count):
    """hammersley.cpp:218-222, 268-270: pixel x owns a contiguous block of the tile's indices (dimension 0), the bit-reversed row
    number in their low digits puts the van der Corput dimension into row y"""
    rng = np.random.RandomState(4)
    for _ in range(40):
        x, y = int(rng.randint(res[0])), int(rng.randint(res[1]))
        for faure in (False, True):
            v = _lib(res, (x, y), faure, 0, 0, array_n=count, n_arrays=HAM, count=count)
            assert (v >= 0).all() and (v < 1).all(), (x, y, v)
    v = _lib(res, (res[0] // 2, res[1] // 2), False, 0, 0, array_n=count, n_arrays=HAM, count=count)
    if count & (count - 1) == 0:
        assert sorted(np.floor(v[:, 0] * count).astype(int)) == list(range(count))      # dimension 0: one sample per column of width 1 / count
        assert sorted(np.floor(v[:, 1] * count).astype(int)) == list(range(count))      # dimension 1: base 2 in the digits above the row number


def test_hammersley_dimensions_are_index_fraction_and_radical_inverses():
    """a 1 x 1 film: index = sample; dimension 0 = j / count, dimension d = radical inverse in the (d - 1)-th prime base"""
    n = 32
    assert np.allclose(_lib((1, 1), (0, 0), False, 1, 0, array_n=n, n_arrays=HAM, count=n), np.arange(n) / n, rtol=3e-7)
    for d, base in enumerate(PRIMES[:6], start=1):
        for faure in (False, True):
            got = _lib((1, 1), (0, 0), faure, 1, d, array_n=n, n_arrays=HAM, count=n)
            perm = _faure(base) if faure else None
            np.testing.assert_allclose(got, [float(_radical_inverse(base, j, perm)) for j in range(n)], rtol=3e-7)


@pytest.mark.parametrize("args", [((640, 480), (3, 5), False, 0, 0), ((640, 480), (300, 205), True, 0, 2), ((64, 48), (63, 47), True, 1, 7),
                                  ((3840, 2160), (3839, 2159), True, 0, 1021), ((640, 480), (-2, -1), True, 0, 0)])
def test_hammersley_library_and_oracle_hand_out_the_same_points(args):
    res, xy, faure, kind, dim = args
    a = _lib(res, xy, faure, kind, dim, array_n=64, n_arrays=HAM, count=64)
    b = kkoracle.halton(res[0], res[1], xy[0], xy[1], faure, kind, dim, 64, HAM, 64)
    np.testing.assert_array_equal(a, b)
    assert (a >= 0).all() and (a < 1).all()


def test_hammersley_host_mirror():
    assert kkdirect.HammersleySampler().render_args() == dict(sampler=abi.SAMPLER_HAMMERSLEY)
    assert kkdirect.HammersleySampler(sampleCount=9, scramble=0).render_args() == dict(sampler=abi.SAMPLER_HAMMERSLEY_PLAIN)
    with pytest.raises(ValueError):
        kkdirect.HammersleySampler(scramble=3)


@pytest.mark.gpu
@pytest.mark.parametrize("sampler", [abi.SAMPLER_HAMMERSLEY, abi.SAMPLER_HAMMERSLEY_PLAIN])
@pytest.mark.parametrize("integrator,kw", [(abi.INTEGRATOR_KK, {}), (abi.INTEGRATOR_DIRECT, {}), (abi.INTEGRATOR_PATH, dict(max_depth=6, rr_depth=2))])
def test_hammersley_images_match_the_oracle(gpu, sampler, integrator, kw):
    sc = _cornell()
    args = dict(integrator=integrator, sampler=sampler, **kw)
    want, _ = kkoracle.Oracle(sc).render(spp=16, seed=4, **args)
    with gpu.Context(device=0, seed=4).upload(sc) as ctx:
        got, _ = ctx.render(16, **args)
        if integrator == abi.INTEGRATOR_KK:
            with pytest.raises(Exception, match="Not supported for the Hammersley"):     # hammersley.cpp:299-303
                ctx.render(4, light_samples=2, sampler=sampler)
    assert np.isfinite(got).all()
    assert helpers.rel_mse(helpers.develop(got), helpers.develop(want)) < 1e-5


@pytest.mark.gpu
def test_hammersley_is_unbiased_and_splits(gpu):
    sc = _cornell(48, 48)
    with gpu.Context(device=0, seed=3).upload(sc) as ctx:
        ref = helpers.develop(ctx.render(1024)[0])
        ind = helpers.develop(ctx.render(16, sample_first=4096)[0])
        whole, _ = ctx.render(16, sampler=abi.SAMPLER_HAMMERSLEY)
        a, _ = ctx.render(8, sampler=abi.SAMPLER_HAMMERSLEY, sample_first=0, sample_count_total=16)
        b, _ = ctx.render(8, sampler=abi.SAMPLER_HAMMERSLEY, sample_first=8, sample_count_total=16)
    ham = helpers.develop(whole)
    np.testing.assert_allclose(a + b, whole, rtol=1e-5, atol=1e-6)
    assert abs(ham.mean() - ref.mean()) < 0.02 * ref.mean()
    assert np.sqrt(((ham - ref) ** 2).mean()) < 0.9 * np.sqrt(((ind - ref) ** 2).mean())
