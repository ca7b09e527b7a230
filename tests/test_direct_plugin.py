"""The stock 'direct' integrator (MIDirectIntegrator, reference src/integrators/direct/direct.cpp:91-334) as a sibling of the kk
integrator on the same kernels.  CPU: the oracle's restatement and the host mirror.  GPU: same-stream
parity of the HIP path against the oracle, through the C ABI (integrator = KKD_INTEGRATOR_DIRECT)."""
import numpy as np
import pytest
import kkoracle
import kkdirect
from kkdirect import abi, scenes
import helpers

DIRECT = abi.INTEGRATOR_DIRECT


def test_host_mirror_parameters_and_serialisation():
    d = kkdirect.MIDirectIntegrator({"shadingSamples": 4})
    assert (d.emitterSamples, d.bsdfSamples, d.strictNormals, d.hideEmitters) == (4, 4, False, False)      # direct.cpp:94-105
    d = kkdirect.MIDirectIntegrator(shadingSamples=4, bsdfSamples=2, strictNormals=True)
    assert (d.emitterSamples, d.bsdfSamples, d.strictNormals) == (4, 2, True)
    e = kkdirect.MIDirectIntegrator.unserialize(d.serialize())
    assert (e.emitterSamples, e.bsdfSamples, e.strictNormals, e.hideEmitters) == (4, 2, True, False)
    assert len(d.serialize()) == 18                                                                    # size, size, bool, bool
    with pytest.raises(ValueError):
        kkdirect.MIDirectIntegrator(emitterSamples=0, bsdfSamples=0)                                   # Assert(sum > 0), direct.cpp:107
    with pytest.raises(ValueError):
        kkdirect.MIDirectIntegrator(lightSamples=1)                                                    # the kk plugin's name for it
    assert kkdirect.MIDirectIntegrator.PLUGIN_NAME == "direct"


def test_oracle_direct_and_kk_estimate_the_same_image():
    """Both integrators are unbiased estimators of direct illumination (balance vs power heuristic); on the Cornell box
    (flat normals: the kk integrator's unconditional geometric-normal test changes nothing) they converge to the same image."""
    sc = scenes.cornell_box(32, 32)
    o = kkoracle.Oracle(sc)
    kk, _ = o.render(512, seed=3)
    st, _ = o.render(512, seed=5, integrator=DIRECT)
    assert helpers.rel_mse(helpers.develop(st), helpers.develop(kk)) < 3e-3
    # one strategy alone is still the same integral (no MIS partner: weight 1)
    em, _ = o.render(1024, 0, 1, seed=7, integrator=DIRECT)
    assert helpers.rel_mse(helpers.develop(em), helpers.develop(kk)) < 3e-3


def test_oracle_strict_normals_only_removes_energy():
    sc = scenes.sponza_class(48, 27, target_tris=4000)
    o = kkoracle.Oracle(sc)
    a, _ = o.render(16, seed=2, integrator=DIRECT)
    b, _ = o.render(16, seed=2, integrator=DIRECT, strict_normals=True)
    assert np.array_equal(a[..., 4], b[..., 4])
    assert b[..., :3].sum() <= a[..., :3].sum() * (1 + 1e-6)


def _same_seed(gpu, sc, spp, tol, **kw):
    o = kkoracle.Oracle(sc)
    ref, so = o.render(spp, seed=1, integrator=DIRECT, **kw)
    with gpu.Context(device=0, seed=1).upload(sc) as ctx:
        film, sg = ctx.render(spp, integrator=DIRECT, **kw)
    assert np.isfinite(film).all() and (film >= 0).all() and sg.samples == so.samples
    r = helpers.rel_mse(helpers.develop(film), helpers.develop(ref))
    assert r < tol, r
    return film, ref, sg, so


@pytest.mark.gpu
def test_direct_plugin_cornell_same_stream(gpu):
    film, ref, sg, so = _same_seed(gpu, scenes.cornell_box(96, 96), 32, 1e-5)
    assert sg.rays_closest == so.rays_closest
    kk, _ = kkoracle.Oracle(scenes.cornell_box(96, 96)).render(32, seed=1)
    assert helpers.rel_mse(helpers.develop(film), helpers.develop(kk)) > 1e-5           # it really is the other estimator


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(bsdf_samples=2, light_samples=3), dict(strict_normals=True), dict(bsdf_samples=3, light_samples=1, strict_normals=True),
                                dict(bsdf_samples=0, light_samples=2), dict(bsdf_samples=2, light_samples=0), dict(hide_emitters=True)])
def test_direct_plugin_all_materials_same_stream(gpu, kw):
    from test_gpu_parity import _material_scene
    _same_seed(gpu, _material_scene(64, 48), 16, 2e-4, **kw)


@pytest.mark.gpu
def test_direct_plugin_host_mirror_renders(gpu):
    sc = scenes.cornell_box(48, 48)
    film, st = kkdirect.MIDirectIntegrator(shadingSamples=2).render(sc, 8)
    ref, _ = kkoracle.Oracle(sc).render(8, 2, 2, seed=1, integrator=DIRECT)
    assert helpers.rel_mse(helpers.develop(film), helpers.develop(ref)) < 1e-5
    with gpu.Context(device=0).upload(sc) as ctx:
        with pytest.raises(gpu.KKDError):
            ctx.render(4, 0, 0, integrator=DIRECT)
        with pytest.raises(gpu.KKDError):
            ctx.render(4, integrator=7)
