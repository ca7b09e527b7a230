"""Independent pins of the GPU path (run with -m gpu): closed-form radiometry through the C ABI, WITHOUT the oracle.

The oracle and the device code restate the same reference files, so their agreement (the same-stream tests) cannot catch
a common misreading of those files.  These tests compare the GPU images with answers that follow from radiometry alone:
white furnace, the form factor of a square light, the inverse-square law, a numerical quadrature of the environment
integral done here in numpy, invariance of the expectation under the sample-count split of the MIS (kk/direct.cpp:162-178
weights with the count fractions), and the kk integrator against its sibling with the other heuristic.
Acceptance level: 1e-2 relative, the single-precision ERROR_REQ of the reference's own sampling tests
(reference src/tests/test_chisquare.cpp:30-38); sample counts are chosen so that Monte Carlo noise is well below it.
"""
import math
import numpy as np
import pytest
from kkdirect import Scene, abi, look_at, scenes, compute_vertex_normals
import helpers

pytestmark = pytest.mark.gpu
ERROR_REQ = 1e-2
PLANE = ([[-60, 0, -60], [-60, 0, 60], [60, 0, 60], [60, 0, -60]], [[0, 1, 2], [0, 2, 3]])       # faces +y


def _mean_rgb(gpu, sc, spp, seed=5, **kw):
    with gpu.Context(device=0, seed=seed).upload(sc) as ctx:
        film, st = ctx.render(spp, **kw)
    assert np.isfinite(film).all()
    return helpers.develop(film)[2:-2, 2:-2].reshape(-1, 3).mean(0)


def _plane_scene(material_fn, eye=(0.0, 3.0, -0.02), fov=8.0, size=16):
    sc = Scene()
    sc.add_mesh(PLANE[0], PLANE[1], material_fn(sc))
    sc.set_sensor(look_at(eye, (0, 0, 0), (0, 1, 0)), fov, size, size, near=0.01, far=1000.0)
    return sc


@pytest.mark.parametrize("counts", [(1, 1), (4, 1), (1, 4)])
def test_white_furnace_diffuse_plane(gpu, counts):
    """constant environment L, unoccluded diffuse plane: Lo = rho * L for every split of the samples between the two
    strategies (diffuse.cpp:110-150 eval / sample / pdf, envmap.cpp sampleDirect / pdfDirect / evalEnvironment)"""
    rho, L = np.array([0.6, 0.45, 0.3]), np.array([1.5, 1.0, 0.5])
    sc = _plane_scene(lambda s: s.diffuse(rho))
    sc.set_envmap(np.tile(L.astype(np.float32), (8, 16, 1)))
    got = _mean_rgb(gpu, sc, 2048, bsdf_samples=counts[0], light_samples=counts[1])
    np.testing.assert_allclose(got, rho * L, rtol=ERROR_REQ)


@pytest.mark.parametrize("make", [lambda s: s.roughconductor((0.143, 0.375, 1.442), (3.983, 2.386, 1.603), alpha=0.3, distribution=abi.DISTR_GGX),
                                  lambda s: s.roughconductor((0.2, 0.92, 1.1), (3.9, 2.45, 2.14), alpha=0.25, alpha_v=0.1),
                                  lambda s: s.roughplastic((0.5, 0.3, 0.2), alpha=0.2),
                                  lambda s: s.roughplastic((0.2, 0.4, 0.6), alpha=0.3, distribution=abi.DISTR_GGX, nonlinear=True)],
                         ids=["conductor-ggx", "conductor-beckmann-aniso", "plastic-beckmann", "plastic-ggx-nonlinear"])
def test_furnace_albedo_is_invariant_under_the_mis_split(gpu, make):
    """rough BSDFs under a constant environment: the albedo has no closed form, but it cannot exceed the environment's
    radiance (energy conservation) and cannot depend on how the samples are split between BSDF and emitter sampling --
    which holds only if eval, pdf and sample of the BSDF and of the emitter are mutually consistent on the device"""
    sc = _plane_scene(make, eye=(0.0, 2.0, -2.5))
    sc.set_envmap(np.ones((8, 16, 3), np.float32))
    got = [_mean_rgb(gpu, sc, 4096, bsdf_samples=b, light_samples=l) for b, l in ((1, 1), (4, 1), (1, 4))]
    assert (got[0] <= 1.0 + ERROR_REQ).all() and (got[0] > 0.05).all()
    for g in got[1:]:
        np.testing.assert_allclose(g, got[0], rtol=ERROR_REQ)


def test_square_light_form_factor(gpu):
    """diffuse plane under a parallel square area light: Lo = rho * L * F, F = the differential-to-rectangle form factor
    (four corner terms) -- area.cpp:104-109,158-183, shape.cpp:99-123, triangle.cpp:24-67, the emitter CDF and both pdfs"""
    rho, L, s, h = 0.7, np.array([9.0, 6.0, 3.0]), 2.0, 1.5
    sc = _plane_scene(lambda sc_: sc_.diffuse((rho,) * 3), eye=(4.0, 1.0, 0.0), fov=1.5, size=12)
    a = s / 2
    sc.add_mesh([[-a, h, -a], [a, h, -a], [a, h, a], [-a, h, a]], [[0, 1, 2], [0, 2, 3]], sc.diffuse((0, 0, 0)), radiance=L)   # faces -y
    def corner(a_, b_, h_):
        A, B = math.sqrt(a_ * a_ + h_ * h_), math.sqrt(b_ * b_ + h_ * h_)
        return (a_ / A * math.atan(b_ / A) + b_ / B * math.atan(a_ / B)) / (2 * math.pi)
    F = 4 * corner(a, a, h)
    for counts in ((1, 1), (3, 1), (1, 3)):
        got = _mean_rgb(gpu, sc, 4096, bsdf_samples=counts[0], light_samples=counts[1])
        np.testing.assert_allclose(got, rho * L * F, rtol=ERROR_REQ)


def test_point_light_inverse_square(gpu):
    """point.cpp:131-151: Lo = I / d^2 * rho / pi * cos(theta); not on a surface, so the MIS weight is 1 (kk/direct.cpp:152)"""
    rho, I = 0.5, np.array([40.0, 30.0, 20.0])
    for pos in ((0.0, 2.0, 0.0), (1.5, 2.0, -1.0)):
        sc = _plane_scene(lambda sc_: sc_.diffuse((rho,) * 3), eye=(3.0, 1.5, 0.5), fov=1.0, size=12)
        sc.add_point_light(pos, I)
        d2 = sum(p * p for p in pos)
        want = I / d2 * rho / math.pi * (pos[1] / math.sqrt(d2))
        np.testing.assert_allclose(_mean_rgb(gpu, sc, 256), want, rtol=ERROR_REQ)


def test_environment_integral_by_quadrature(gpu):
    """a non-constant environment map on an unoccluded diffuse plane: Lo = rho / pi * integral of L cos over the upper
    hemisphere, the integral done here by midpoint quadrature over the bilinearly reconstructed lat-long image
    (envmap.cpp:380-410 lookup convention: uv = (atan2(x, -z) / 2 pi, acos(y) / pi), texel centres at +0.5, repeat in u,
    clamp in v) -- pins evalEnvironment, the two-level sampling table and its pdf against plain numpy"""
    rng = np.random.RandomState(11)
    H, W = 16, 32
    tex = (0.2 + rng.rand(H, W, 3) ** 2).astype(np.float32)
    tex[3, 9] = (30.0, 25.0, 20.0); tex[5, 22] = (4.0, 8.0, 12.0)        # two bright texels above the horizon
    tex = np.float16(tex).astype(np.float32)                               # the library stores half-float texels (envmap.cpp:102-103)
    rho = np.array([0.8, 0.6, 0.4])
    sc = _plane_scene(lambda s: s.diffuse(rho))
    sc.set_envmap(tex)
    S = 16                                                                 # quadrature cells per texel and axis
    v = (np.arange(H * S) + 0.5) / (H * S); u = (np.arange(W * S) + 0.5) / (W * S)
    y = v * H - 0.5; x = u * W - 0.5
    y0 = np.floor(y).astype(int); x0 = np.floor(x).astype(int); fy = (y - y0)[:, None, None]; fx = (x - x0)[None, :, None]
    ya, yb = np.clip(y0, 0, H - 1), np.clip(y0 + 1, 0, H - 1); xa, xb = x0 % W, (x0 + 1) % W
    img = (tex[ya][:, xa] * (1 - fx) + tex[ya][:, xb] * fx) * (1 - fy) + (tex[yb][:, xa] * (1 - fx) + tex[yb][:, xb] * fx) * fy
    theta = v * math.pi
    w = np.clip(np.cos(theta), 0, None) * np.sin(theta) * (math.pi / (H * S)) * (2 * math.pi / (W * S))   # cos * d omega, upper hemisphere
    want = rho / math.pi * (img * w[:, None, None]).sum((0, 1))
    for counts in ((1, 1), (1, 4)):
        got = _mean_rgb(gpu, sc, 8192, bsdf_samples=counts[0], light_samples=counts[1])
        np.testing.assert_allclose(got, want, rtol=ERROR_REQ)


def test_kk_and_stock_direct_converge_to_the_same_image(gpu):
    """balance heuristic with count fractions (kk/direct.cpp:162-178) vs power heuristic (direct/direct.cpp:311-314): two
    unbiased estimators of the same integral (strictNormals on, so both apply the geometric-normal test of kk/direct.cpp:
    98,154) -- their converged images agree within the north-star bar, relMSE <= 1e-3"""
    from test_gpu_parity import _material_scene
    sc = _material_scene(64, 48)
    with gpu.Context(device=0, seed=21).upload(sc) as ctx:
        a, _ = ctx.render(4096, 1, 1)
        b, _ = ctx.render(4096, 1, 1, integrator=abi.INTEGRATOR_DIRECT, strict_normals=True)
        c, _ = ctx.render(2048, 2, 2, sample_first=10_000)
    assert helpers.rel_mse(helpers.develop(a), helpers.develop(b)) <= 1e-3
    assert helpers.rel_mse(helpers.develop(c), helpers.develop(a)) <= 1e-3


def _emitting_room(rho, Le, size=12):
    """camera inside a closed box whose six walls are diffuse (rho) AND emit Le (faces point inwards)"""
    sc = Scene()
    v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]     # outward winding
    idx = [t for a, b, c, d in quads for t in ((a, c, b), (a, d, c))]                                 # flipped: normals point in
    sc.add_mesh(v, idx, sc.diffuse((rho,) * 3), radiance=Le)
    sc.set_sensor(look_at((0.1, -0.2, 0.3), (1, 0.4, -0.2), (0, 1, 0)), 50.0, size, size, near=0.01, far=100.0)
    return sc


def test_emitting_enclosure_series(gpu):
    """a closed enclosure with uniform emission Le and albedo rho has the uniform solution Le / (1 - rho) whatever its shape;
    cut at n vertices it is Le (1 + rho + ... + rho^(n-1)).  kk and stock direct = two terms, path with maxDepth n = n terms, path
    without a bound (Russian roulette from depth 3) = the limit.  Pins the area emitter (sampling, pdf, eval on hit), the
    diffuse BSDF, both MIS weights, the depth rule of path.cpp:135-165 and the roulette -- without the oracle"""
    rho, Le = 0.6, np.array([2.0, 1.0, 0.5])
    sc = _emitting_room(rho, Le)
    np.testing.assert_allclose(_mean_rgb(gpu, sc, 4096), Le * (1 + rho), rtol=ERROR_REQ)
    np.testing.assert_allclose(_mean_rgb(gpu, sc, 2048, bsdf_samples=2, light_samples=3), Le * (1 + rho), rtol=ERROR_REQ)
    np.testing.assert_allclose(_mean_rgb(gpu, sc, 4096, integrator=abi.INTEGRATOR_DIRECT), Le * (1 + rho), rtol=ERROR_REQ)
    for n in (1, 2, 3, 6):
        want = Le * sum(rho ** k for k in range(n))
        np.testing.assert_allclose(_mean_rgb(gpu, sc, 4096, integrator=abi.INTEGRATOR_PATH, max_depth=n, rr_depth=100), want, rtol=ERROR_REQ)
    np.testing.assert_allclose(_mean_rgb(gpu, sc, 8192, integrator=abi.INTEGRATOR_PATH, max_depth=-1, rr_depth=3), Le / (1 - rho), rtol=ERROR_REQ)
    # hideEmitters removes exactly the directly visible term
    np.testing.assert_allclose(_mean_rgb(gpu, sc, 4096, hide_emitters=True), Le * rho, rtol=ERROR_REQ)


def test_glass_sphere_in_a_furnace(gpu):
    """a non-absorbing dielectric under a constant environment L, path plugin without a depth bound: every path leaves the
    sphere with throughput 1 (Fresnel-sampled lobes have weight 1, the radiance scaling (1 / eta)^2 on entry cancels eta^2 on
    exit, dielectric.cpp:217-333), so EVERY pixel is L whatever the sample count -- no Monte Carlo noise at all without
    roulette.  Pins the two delta lobes, the eta bookkeeping, total internal reflection and the tracer on a closed mesh."""
    from kkdirect.scenes import cube_sphere
    L = np.array([1.25, 0.75, 0.5], np.float32)
    sc = Scene()
    p, i = cube_sphere(24, 1.0, 0.0, 1)
    sc.add_mesh(p, i, sc.dielectric(1.5), normals=compute_vertex_normals(p, i))
    sc.set_envmap(np.tile(L, (8, 16, 1)))
    sc.set_sensor(look_at((0, 0.3, -4), (0, 0, 0), (0, 1, 0)), 35.0, 32, 32, near=0.01, far=100.0)
    with gpu.Context(device=0, seed=5).upload(sc) as ctx:
        film, _ = ctx.render(8, integrator=abi.INTEGRATOR_PATH, max_depth=-1, rr_depth=100000)
    img = helpers.develop(film)[2:-2, 2:-2].reshape(-1, 3)
    # (a ray in a thousand runs along the faceted surface until the 4096-vertex bound of the path kernel cuts it)
    assert (np.abs(img / L - 1) < 2e-3).mean() > 0.99 and abs(img.mean(0) / L - 1).max() < 2e-3


def _conductor_fresnel(c, eta, k):
    """exact Fresnel reflectance of a conductor for unpolarised light (textbook form, written out here independently)"""
    c2 = c * c; s2 = 1 - c2; s4 = s2 * s2
    t1 = eta * eta - k * k - s2
    a2pb2 = np.sqrt(t1 * t1 + 4 * k * k * eta * eta)
    a = np.sqrt(0.5 * (a2pb2 + t1))
    term1 = a2pb2 + c2; term2 = 2 * a * c
    rs = (term1 - term2) / (term1 + term2)
    term3 = a2pb2 * c2 + s4; term4 = term2 * s2
    return 0.5 * (rs + rs * (term3 - term4) / (term3 + term4))


def _microfacet_d_g1(distr, alpha):
    if distr == abi.DISTR_BECKMANN:
        def D(ct):
            t2 = (1 - ct * ct) / (ct * ct)
            return math.exp(-t2 / alpha ** 2) / (math.pi * alpha ** 2 * ct ** 4)
        def G1(ct):
            tan = math.sqrt(1 - ct * ct) / ct
            if tan == 0: return 1.0
            a = 1 / (alpha * tan)
            return 1.0 if a >= 1.6 else (3.535 * a + 2.181 * a * a) / (1 + 2.276 * a + 2.577 * a * a)
    else:
        def D(ct):
            t2 = (1 - ct * ct) / (ct * ct)
            return 1 / (math.pi * alpha ** 2 * ct ** 4 * (1 + t2 / alpha ** 2) ** 2)
        def G1(ct):
            t2 = (1 - ct * ct) / (ct * ct)
            return 2 / (1 + math.sqrt(1 + alpha * alpha * t2))
    return D, G1


@pytest.mark.parametrize("distr,alpha", [(abi.DISTR_BECKMANN, 0.3), (abi.DISTR_GGX, 0.25), (abi.DISTR_BECKMANN, 0.15)])
@pytest.mark.parametrize("light_dir", [(0.5, 1.0, 0.2), (-0.3, 1.0, 0.9)])      # off-specular and near the mirror direction
def test_rough_conductor_under_a_point_light(gpu, distr, alpha, light_dir):
    """one point light, one viewing direction: Lo = F(wi.h) D(h) G1(wi) G1(wo) / (4 cos theta_view) * I / d^2 with h the
    half vector -- the microfacet model of roughconductor.cpp:257-300 / microfacet.h evaluated here in plain Python from the
    textbook formulas (Beckmann / GGX D, Smith G1 with Walter's rational fit for Beckmann, exact conductor Fresnel), not through
    the oracle.  A point light has MIS weight 1 (kk/direct.cpp:152) and BSDF samples cannot reach it: the value is exact."""
    eta = np.array([0.2, 0.92, 1.1]); k = np.array([3.9, 2.45, 2.14]); I = np.array([30.0, 30.0, 30.0]); ext = 1.000277
    eye = np.array([0.0, 2.0, -2.5])
    sc = _plane_scene(lambda s: s.roughconductor(eta, k, alpha=alpha, distribution=distr), eye=tuple(eye), fov=0.5, size=8)
    lp = np.array(light_dir, float) * 3.0
    sc.add_point_light(tuple(lp), I)
    wv = eye / np.linalg.norm(eye); wl = lp / np.linalg.norm(lp)
    h = wv + wl; h /= np.linalg.norm(h)
    D, G1 = _microfacet_d_g1(distr, alpha)
    F = _conductor_fresnel(float(wv @ h), eta / ext, k / ext)
    want = F * D(h[1]) * G1(wv[1]) * G1(wl[1]) / (4 * wv[1]) * I / float(lp @ lp)
    got = _mean_rgb(gpu, sc, 64)
    np.testing.assert_allclose(got, want, rtol=ERROR_REQ)


@pytest.mark.parametrize("distr,alpha,nonlinear", [(abi.DISTR_BECKMANN, 0.2, False), (abi.DISTR_GGX, 0.3, True)])
@pytest.mark.parametrize("light_dir", [(0.5, 1.0, 0.2), (-0.3, 1.0, 0.9)])
def test_rough_plastic_under_a_point_light(gpu, distr, alpha, nonlinear, light_dir):
    """roughplastic.cpp:340-380 under one point light, written out from the model, not through the oracle:
        Lo = [ Ks F(wi.h) D(h) G1(wi) G1(wo) / (4 cos_i)  +  Kd' / pi * cos_o * T12(cos_i) T21(cos_o) / eta^2 ] * I / d^2,
    Kd' = Kd / (1 - Fdr) or Kd / (1 - Kd Fdr) ('nonlinear'), Fdr = 1 - the diffuse transmittance seen from inside.  F is the
    textbook dielectric Fresnel term; T12, T21 and Fdr come from the independent numpy integration of the rough dielectric
    boundary in tests/test_rtrans_protocol.py (the reference's own protocol for that table), so this pins the plastic's eval on the
    device -- specular lobe, diffuse lobe and the table lookups -- by something that shares no code with product or oracle."""
    import test_rtrans_protocol as rt
    int_ior, ext = 1.49, 1.000277
    eta = float(np.float32(int_ior) / np.float32(ext))
    Kd = np.array([0.6, 0.3, 0.1]); Ks = 1.0; I = np.array([30.0, 30.0, 30.0])
    eye = np.array([0.0, 2.0, -2.5])
    sc = _plane_scene(lambda s: s.roughplastic(tuple(Kd), alpha=alpha, int_ior=int_ior, ext_ior=ext, distribution=distr, nonlinear=nonlinear),
                      eye=tuple(eye), fov=0.5, size=8)
    lp = np.array(light_dir, float) * 3.0
    sc.add_point_light(tuple(lp), I)
    wv = eye / np.linalg.norm(eye); wl = lp / np.linalg.norm(lp)
    h = wv + wl; h /= np.linalg.norm(h)
    D, G1 = _microfacet_d_g1(distr, alpha)
    F = float(rt.fresnel_dielectric(np.array(float(wv @ h)), eta)[0])
    spec = Ks * F * D(h[1]) * G1(wv[1]) * G1(wl[1]) / (4 * wv[1])
    T12 = rt.transmittance(distr, alpha, eta, float(wv[1])); T21 = rt.transmittance(distr, alpha, eta, float(wl[1]))
    x, w = np.polynomial.legendre.leggauss(24)
    mu, w = 0.5 * (x + 1), 0.5 * w
    fdr = 1.0 - float(sum(wk * 2 * m * rt.transmittance(distr, alpha, 1.0 / eta, float(m), 192, 192) for m, wk in zip(mu, w)))
    kd = Kd / (1 - Kd * fdr) if nonlinear else Kd / (1 - fdr)
    want = (spec + kd / math.pi * wl[1] * T12 * T21 / (eta * eta)) * I / float(lp @ lp)
    got = _mean_rgb(gpu, sc, 64)
    np.testing.assert_allclose(got, want, rtol=ERROR_REQ)
    if light_dir[2] > 0.5:      # near the mirror direction the specular lobe is a visible part of the answer, off it the diffuse lobe is alone
        assert spec * I[0] / float(lp @ lp) > 0.02 * want.min()
