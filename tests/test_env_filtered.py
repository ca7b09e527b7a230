"""The filtered environment lookup of camera rays (reference src/emitters/envmap.cpp:379-409: ray differentials -> texture-space
footprint -> TMIPMap::eval with the EWA filter; perspective.cpp:294-295 and integrator.cpp:144-145, 181 for the differentials).

TMIPMap / Resampler live in headers the reference checkout does not hold (mipmap.h, rfilter.h): both sides restate the
published Mitsuba 0.5 sources, parity against the reference itself is unpinned.  What is tested: the oracle's pyramid and
filter against properties that hold for any correct implementation, the device against the oracle on the same streams,
and the device alone against the closed form (a footprint that covers many texels returns their mean)."""
import numpy as np
import pytest
import kkoracle
from kkdirect import Scene, abi, look_at
import helpers


def _env_only_scene(tex, size=32, fov=90.0, target=(0.3, 0.2, -1.0), rfilter=None):
    sc = Scene()
    # one small triangle behind the camera: the scene needs geometry, every camera ray leaves it
    sc.add_mesh([[-0.1, -0.1, 5.0], [0.1, -0.1, 5.0], [0.0, 0.1, 5.0]], [[0, 1, 2]], sc.diffuse((0.5, 0.5, 0.5)))
    sc.set_envmap(np.ascontiguousarray(tex, np.float32))
    kw = {} if rfilter is None else dict(rfilter=rfilter)
    sc.set_sensor(look_at((0, 0, 0), target, (0, 1, 0)), fov, size, size, near=0.01, far=100.0, **kw)
    return sc


def _stripes(h, w, lo=0.0, hi=2.0):
    tex = np.empty((h, w, 3), np.float32)
    tex[:, 0::2] = lo; tex[:, 1::2] = hi                  # one texel wide, alternating along u: mean (lo + hi) / 2
    return tex


def test_oracle_filtered_lookup_properties():
    """a constant map stays constant through the pyramid and the filter; one-texel stripes minified 16x come back as their
    mean where bilinear level-0 sampling returns anything in [lo, hi]"""
    const = np.tile(np.array([0.7, 0.4, 0.2], np.float32), (64, 128, 1))
    film, _ = kkoracle.Oracle(_env_only_scene(const, size=16)).render(spp=4, seed=1)
    img = helpers.develop(film)
    np.testing.assert_allclose(img.reshape(-1, 3), np.tile([0.7, 0.4, 0.2], (img.shape[0] * img.shape[1], 1)), rtol=2e-3)
    film, _ = kkoracle.Oracle(_env_only_scene(_stripes(256, 512), size=16)).render(spp=1, seed=1)   # 512 texels over 360 deg, a pixel spans ~8
    img = helpers.develop(film)[..., 0]
    assert abs(img.mean() - 1.0) < 0.02 and img.std() < 0.05, (img.mean(), img.std())


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["minified-noise", "magnified", "anisotropic-near-pole", "tiny-map"])
def test_filtered_lookup_matches_the_oracle(gpu, case):
    rng = np.random.RandomState(3)
    if case == "minified-noise":        # EWA over several texels, two levels blended
        tex, size, fov, target, spp = rng.rand(512, 1024, 3).astype(np.float32) * 2, 48, 80.0, (0.3, 0.2, -1.0), 4
    elif case == "magnified":           # footprint below one texel: bilinear level 0 (the bulk of C1-C4)
        tex, size, fov, target, spp = rng.rand(32, 64, 3).astype(np.float32), 64, 40.0, (0.3, 0.2, -1.0), 2
    elif case == "anisotropic-near-pole":   # u compressed near the pole: long ellipses, the anisotropy clamp
        tex, size, fov, target, spp = rng.rand(256, 512, 3).astype(np.float32), 48, 60.0, (0.05, 1.0, -0.15), 1
    else:                               # levels run out: evalBox on the 1 x 1 level
        tex, size, fov, target, spp = rng.rand(2, 4, 3).astype(np.float32), 8, 150.0, (0.0, 0.1, -1.0), 1
    sc = _env_only_scene(tex, size=size, fov=fov, target=target)
    want, _ = kkoracle.Oracle(sc).render(spp=spp, seed=9)
    with gpu.Context(device=0, seed=9).upload(sc) as ctx:
        got, _ = ctx.render(spp)
    assert np.isfinite(got).all()
    # same streams, same restatement: what is left is the device's reciprocal-multiply divisions in the footprint arithmetic
    assert helpers.rel_mse(helpers.develop(got), helpers.develop(want)) < 1e-5


@pytest.mark.gpu
def test_minified_stripes_average_on_the_device(gpu):
    """closed form, no oracle: one-texel stripes of 0 and 2 seen through pixels that span ~8 texels each -> 1 everywhere;
    unfiltered point sampling would give values between 0 and 2 (std ~0.7 at one sample per pixel)"""
    sc = _env_only_scene(_stripes(256, 512), size=16)
    with gpu.Context(device=0, seed=2).upload(sc) as ctx:
        film, _ = ctx.render(1)
    img = helpers.develop(film)[..., 0]
    assert abs(img.mean() - 1.0) < 0.02 and img.std() < 0.05, (img.mean(), img.std())
    # and a constant map is returned unchanged at any footprint
    sc = _env_only_scene(np.tile(np.array([0.7, 0.4, 0.2], np.float32), (64, 128, 1)), size=16)
    with gpu.Context(device=0, seed=2).upload(sc) as ctx:
        film, _ = ctx.render(4)
    img = helpers.develop(film).reshape(-1, 3)
    np.testing.assert_allclose(img, np.tile([0.7, 0.4, 0.2], (len(img), 1)), rtol=2e-3)


@pytest.mark.gpu
def test_pyramid_levels_are_aligned_with_level_zero(gpu):
    """closed form, no oracle: a smooth map (three periods around, two from pole to pole) looks the same through the pyramid as
    through level 0.  At 1024 samples per pixel the ray differentials are pulled in by 1 / sqrt(spp) (integrator.cpp:144-145),
    the footprint is below a texel, every lookup is the bilinear level-0 value and the pixel filter does the averaging; at ONE
    sample per pixel a footprint spans ~16 texels and the answer comes from levels 3-5 through the EWA filter (48 such renders
    with different seeds are averaged, so that the position of the one sample inside its pixel averages out: +-0.03 -> +-0.004).
    A resampler whose pixel centres were off by half a texel on some level (the classic mistake in (i + 0.5) / target * source)
    would shift the pattern there by 2^level / 2 texels: at level 4 that is 4 degrees of phase, 0.03 in these units, everywhere
    the pattern has a slope -- the bound is 0.012."""
    h, w = 1024, 2048
    u = (np.arange(w) + 0.5) / w; v = (np.arange(h) + 0.5) / h
    lum = 1.0 + 0.4 * np.sin(2 * np.pi * 3 * u)[None, :] + 0.3 * np.cos(2 * np.pi * 2 * v)[:, None]
    tex = np.repeat(lum[..., None], 3, axis=2).astype(np.float32)
    sc = _env_only_scene(tex, size=32, fov=90.0, target=(0.3, 0.2, -1.0))
    coarse = 0.0
    with gpu.Context(device=0, seed=4).upload(sc) as ctx:
        fine, _ = ctx.render(1024)
    n = 48
    for seed in range(n):
        with gpu.Context(device=0, seed=100 + seed).upload(sc) as ctx:
            film, _ = ctx.render(1)
        coarse = coarse + film.astype(np.float64)
    a, b = helpers.develop(fine)[2:-2, 2:-2, 0], helpers.develop(coarse.astype(np.float32))[2:-2, 2:-2, 0]
    assert a.std() > 0.1                                    # the pattern is in view
    d = np.abs(b - a)
    assert d.max() < 0.012 and d.mean() < 0.004, (d.max(), d.mean())
