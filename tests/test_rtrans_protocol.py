"""The rough-transmittance table of roughplastic, pinned by the reference's own acceptance protocol.

`data/microfacet/*.dat` is not in the checkout, so library (kkd_host.cpp: roughTransmittanceTable) and oracle re-integrate
the table.  Both were written by the same hand from the same formula; this file shares no code with either.  It restates, in
plain numpy, what `src/tests/test_rtrans.cpp:28-36, 94-124, 141-196` does: the transmittance of a rough dielectric at cos(theta)
is the mean over [0,1]^2 of the weight `RoughDielectric::sample` returns when only the transmission component is allowed
(`src/bsdfs/roughdielectric.cpp:424-498`: microfacet normal m from the distribution of visible normals, weight = (1 - F(wi.m)) *
G1(wo, m), zero when the refracted direction stays on the side of wi), importance mode (no eta^2 factor), and the table must
agree within 1e-3 ABSOLUTE (`test_rtrans.cpp:158, 184`); the diffuse transmittance is 2 int_0^1 mu T(mu) dmu
(`test_rtrans.cpp:38-41, 84-88`); `fresnelDiffuseReflectance` fast vs exact within 1e-3 for eta = 1 ... 2 and 1 / eta
(`test_rtrans.cpp:126-139`, fits of `src/libcore/util.cpp:788-826`).

The mean over the visible-normal sampler is written as the integral it samples (Heitz & d'Eon 2014: D_wi(m) = G1(wi, m)
max(0, wi.m) D(m) / cos(theta_i)) and evaluated by a midpoint rule in the coordinates (u, phi) in which D(m) cos(theta_m) is
uniform, so the same code serves alpha = 1e-3 ... 4.  CPU only: kkd_probe_rough_transmittance runs on the host.
"""
import ctypes as C
import numpy as np
import pytest
import kkdirect
from kkdirect import abi

BECKMANN, GGX = abi.DISTR_BECKMANN, abi.DISTR_GGX


def fresnel_dielectric(cos_i, eta):
    """unpolarised Fresnel reflectance, textbook form; eta = n_t / n_i for cos_i > 0, inverted for cos_i < 0; also cos_t (signed)"""
    cos_i = np.asarray(cos_i, np.float64)
    scale = np.where(cos_i > 0, 1.0 / eta, eta)
    cos_t2 = 1.0 - (1.0 - cos_i * cos_i) * scale * scale
    tir = cos_t2 <= 0
    ci = np.abs(cos_i)
    ct = np.sqrt(np.where(tir, 0.0, cos_t2))
    e = np.where(cos_i > 0, eta, 1.0 / eta)
    rs = (ci - e * ct) / (ci + e * ct)
    rp = (e * ci - ct) / (e * ci + ct)
    F = np.where(tir, 1.0, 0.5 * (rs * rs + rp * rp))
    return F, np.where(cos_i > 0, -ct, ct)


def smith_g1(v, m, alpha, distr):
    """Smith shadowing of one direction (Walter et al. 2007: rational fit for Beckmann, closed form for GGX)"""
    vz = v[..., 2]
    ok = (np.sum(v * m, -1) * vz) > 0
    tan_v = np.sqrt(np.maximum(0.0, 1 - vz * vz)) / np.maximum(np.abs(vz), 1e-300)
    with np.errstate(divide="ignore", invalid="ignore"):
        if distr == BECKMANN:
            a = 1.0 / (alpha * tan_v)
            g = np.where(a >= 1.6, 1.0, (3.535 * a + 2.181 * a * a) / (1.0 + 2.276 * a + 2.577 * a * a))
        else:
            g = 2.0 / (1.0 + np.sqrt(1.0 + (alpha * tan_v) ** 2))
    g = np.where(tan_v == 0, 1.0, g)
    return np.where(ok, g, 0.0)


def transmittance(distr, alpha, eta, cos_i, n_u=384, n_phi=384):
    """int D_wi(m) (1 - F(wi.m)) G1(wo, m) [wo below the surface] dm, midpoint rule"""
    # graded in u: u = 1 - (1 - v)^4 resolves the tail u -> 1 (steep normals), where a grazing wi sees most of its visible normals
    v = (np.arange(n_u) + 0.5) / n_u
    u, wu = 1 - (1 - v) ** 4, 4 * (1 - v) ** 3
    phi = (np.arange(n_phi) + 0.5) / n_phi * 2 * np.pi
    if distr == BECKMANN:
        tan2 = -alpha * alpha * np.log1p(-u)                 # D cos is uniform in u = 1 - exp(-tan^2 / alpha^2)
    else:
        tan2 = alpha * alpha * u / (1 - u)                   # ... in u = tan^2 / (alpha^2 + tan^2)
    cos_m = 1.0 / np.sqrt(1.0 + tan2)
    sin_m = np.sqrt(np.maximum(0.0, 1 - cos_m * cos_m))
    m = np.stack([sin_m[:, None] * np.cos(phi)[None, :], sin_m[:, None] * np.sin(phi)[None, :], np.broadcast_to(cos_m[:, None], (n_u, n_phi))], -1)
    wi = np.array([np.sqrt(max(0.0, 1 - cos_i * cos_i)), 0.0, cos_i])
    wi_m = m @ wi
    F, cos_t = fresnel_dielectric(wi_m, eta)
    # refract(wi, m, eta, cosThetaT) of util.cpp: m * (wi.m * scale + cosThetaT) - wi * scale, scale = 1 / eta for wi.m > 0
    scale = np.where(cos_t < 0, 1.0 / eta, eta)
    wo = m * (wi_m * scale + cos_t)[..., None] - wi[None, None, :] * scale[..., None]
    side = (cos_i * wo[..., 2] < 0) & (cos_t != 0)
    vis = smith_g1(np.broadcast_to(wi, m.shape), m, alpha, distr) * np.maximum(0.0, wi_m) / (cos_i * cos_m[:, None])
    w = vis * (1 - F) * smith_g1(wo, m, alpha, distr) * side
    return float((w.mean(axis=1) * wu).mean())


def probe(distr, alpha, eta):
    lib = kkdirect.load_library()
    t = np.zeros(100, np.float32); d = np.zeros(1, np.float32)
    assert lib.kkd_probe_rough_transmittance(distr, C.c_float(alpha), C.c_float(eta), t.ctypes.data_as(abi.pf32), d.ctypes.data_as(abi.pf32)) == 0
    return t, float(d[0])


def _draws(n, seed):
    """the parameter distribution of test_rtrans.cpp:145-151, 167-176 (alpha = u^4 * 4, eta = 1 + u^4 * 3), floored where the plugin
    itself clamps (roughplastic.cpp / microfacet.h: alpha >= 1e-4) and capped at alpha = 2, beyond which no material of the path lives"""
    rng = np.random.RandomState(seed)
    for _ in range(n):
        alpha = min(2.0, max(1e-3, rng.rand() ** 4 * 4))
        eta = max(1 + 1e-3, 1 + rng.rand() ** 4 * 3)
        yield float(np.float32(alpha)), float(np.float32(eta)), rng


@pytest.mark.parametrize("distr", [BECKMANN, GGX])
def test_table_against_brute_force_integration(distr):
    worst = 0.0
    for alpha, eta, rng in _draws(60, seed=11 + distr):
        tab, _ = probe(distr, alpha, eta)
        # knots of the table: cos(theta)^(1/4) = i / 99 (rtrans.h, rdielprec.cpp:88-95).  The reference draws cos(theta) uniformly
        # (test_rtrans.cpp:168): 99 % of its draws lie above knot 32 (cos = 0.011); the grazing knots below have a test of their own
        for knot in (int(rng.randint(32, 100)), int(rng.randint(32, 100))):
            cos_i = (knot / 99.0) ** 4
            ref = transmittance(distr, alpha, eta, cos_i)
            worst = max(worst, abs(ref - float(tab[knot])))
            assert abs(ref - float(tab[knot])) <= 1e-3, (distr, alpha, eta, knot, ref, float(tab[knot]))
    print("worst |table - brute force| = %.2e" % worst)


@pytest.mark.parametrize("distr", [BECKMANN, GGX])
def test_grazing_knots(distr):
    """cos(theta) = 1e-4 ... 1e-2: the visible normals crowd into the far tail of the distribution, so the midpoint rule is
    refined until two resolutions agree before the table is held against it"""
    for alpha, eta in ((0.1, 1.49), (0.3, 1.5), (0.05, 1 / 1.49)):
        tab, _ = probe(distr, alpha, eta)
        for knot in (10, 20, 30):
            cos_i = (knot / 99.0) ** 4
            a, b = transmittance(distr, alpha, eta, cos_i, 1024, 512), transmittance(distr, alpha, eta, cos_i, 2048, 512)
            assert abs(a - b) <= 3e-4, (distr, alpha, eta, knot, a, b)
            assert abs(b - float(tab[knot])) <= 1e-3, (distr, alpha, eta, knot, b, float(tab[knot]))


@pytest.mark.parametrize("distr", [BECKMANN, GGX])
def test_both_directions_and_the_materials_of_the_configs(distr):
    """roughplastic needs the table for eta and for 1 / eta (roughplastic.cpp:255-307: external and internal tables)"""
    for alpha, eta in ((0.1, 1.49 / 1.000277), (0.1, 1.000277 / 1.49), (0.3, 1 / 1.5), (0.05, 1.33)):
        tab, _ = probe(distr, alpha, eta)
        for knot in (40, 70, 99):
            ref = transmittance(distr, alpha, eta, (knot / 99.0) ** 4)
            assert abs(ref - float(tab[knot])) <= 1e-3, (distr, alpha, eta, knot, ref, float(tab[knot]))


def test_diffuse_transmittance_against_brute_force_integration():
    x, w = np.polynomial.legendre.leggauss(40)
    mu, w = 0.5 * (x + 1), 0.5 * w
    for alpha, eta, _ in list(_draws(12, seed=5)) + [(0.1, 1.49 / 1.000277, None), (0.1, 1.000277 / 1.49, None)]:
        _, d = probe(BECKMANN, alpha, eta)
        ref = float(sum(wk * 2 * m * transmittance(BECKMANN, alpha, eta, float(m), 192, 192) for m, wk in zip(mu, w)))
        assert abs(ref - d) <= 1e-3, (alpha, eta, ref, d)


def _fdr_fast(eta):
    """fresnelDiffuseReflectance(eta, fast = true), src/libcore/util.cpp:796-826 (Egan & Hilgeman 1973; d'Eon & Irving 2011)"""
    if eta < 1:
        return -1.4399 * eta * eta + 0.7099 * eta + 0.6681 + 0.0636 / eta
    i = 1.0 / eta
    return 0.919317 - 3.4793 * i + 6.75335 * i ** 2 - 7.80989 * i ** 3 + 4.98554 * i ** 4 - 1.36881 * i ** 5


def _fdr_exact(eta):
    """fresnelDiffuseReflectance(eta, fast = false): int_0^1 F(sqrt(xi), eta) dxi (util.cpp:782-786, 827-833)"""
    x, w = np.polynomial.legendre.leggauss(400)
    xi, w = 0.5 * (x + 1), 0.5 * w
    return float((w * fresnel_dielectric(np.sqrt(xi), eta)[0]).sum())


def test_fresnel_diffuse_reflectance_fast_vs_exact_and_the_smooth_limit():
    for i in range(11):                                     # test_rtrans.cpp:126-139
        eta = 1 + i / 10.0
        assert abs(_fdr_fast(eta) - _fdr_exact(eta)) <= 1e-3
        assert abs(_fdr_fast(1 / eta) - _fdr_exact(1 / eta)) <= 1e-3
    # a nearly smooth interface transmits 1 - Fdr diffusely (what rdielprec.cpp:109 prints side by side)
    for eta in (1.2, 1.49, 2.0):
        _, d = probe(BECKMANN, 1e-3, eta)
        assert abs(d - (1 - _fdr_exact(eta))) <= 2e-3, (eta, d, 1 - _fdr_exact(eta))
    # from the dense side T(mu) of a nearly smooth interface drops to zero at the critical angle: the reference integrates the
    # diffuse value through the 100-knot cubic interpolant (test_rtrans.cpp:38-41, rdielprec.cpp:58-62), which cannot follow the
    # step, and so does the library -- the difference to the exact value is that interpolation error, not a table error
    _, d = probe(BECKMANN, 1e-3, 1 / 1.49)
    assert abs(d - (1 - _fdr_exact(1 / 1.49))) <= 1e-2
