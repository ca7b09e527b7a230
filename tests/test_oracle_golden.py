"""CPU tests: the oracle (CPU restatement of the reference path) against the known answers the reference's own
tests hold, plus internal consistency of the restatement (kd-tree vs brute force, sampling vs pdf)."""
import numpy as np
import pytest
import kkoracle
from kkdirect import Scene, abi, scenes
import helpers


def _one_triangle(golden, case):
    g = golden["dgeom"]
    sc = Scene()
    m = sc.diffuse()
    c = g[case]
    sc.add_mesh(g["vertices"], [[0, 1, 2]], m, normals=c["normals"], texcoords=c["uv"])
    sc.set_sensor(np.eye(4), 45, 8, 8)
    ray = np.array(g["ray"]["o"] + [0.0] + g["ray"]["d"] + [np.inf], np.float32)
    return kkoracle.Oracle(sc).intersect_record(ray), c


@pytest.mark.parametrize("case", ["test01", "test02", "test03"])
def test_dgeom_hit_record(golden, case):
    """reference src/tests/test_dgeom.cpp:35-177"""
    rec, c = _one_triangle(golden, case)
    assert rec is not None
    eps = 1e-4   # Epsilon of the reference's assertEqualsEpsilon
    np.testing.assert_allclose(rec["p"], c["p"], atol=1e-6)
    np.testing.assert_allclose(rec["uv"], c["its_uv"], atol=eps)
    np.testing.assert_allclose(rec["geoN"], c["geoN"], atol=1e-6)
    if "shN" in c:
        np.testing.assert_allclose(rec["shN"], c["shN"], atol=1e-6)
    else:
        n = np.asarray(c["normals"], np.float64); w = np.asarray(c["shN_weights"])
        e = (n * w[:, None]).sum(0); e /= np.linalg.norm(e)
        np.testing.assert_allclose(rec["shN"], e, atol=eps)
    np.testing.assert_allclose(rec["dpdu"], c["dpdu"], atol=eps)
    if "dpdv" in c:
        np.testing.assert_allclose(rec["dpdv"], c["dpdv"], atol=eps)
    assert abs(rec["t"] - 1.0) < 1e-6


def test_sutherland_hodgman_clip(golden):
    """reference src/tests/test_kd.cpp:34-84"""
    tri = golden["clip"]["triangle"]
    for c in golden["clip"]["cases"]:
        r = kkoracle.clip_aabb(tri, c["box"])
        if c.get("valid", True):
            np.testing.assert_array_equal(r[0], np.float32(c["min"]))
            np.testing.assert_array_equal(r[1], np.float32(c["max"]))
        else:
            assert not np.all(r[1] >= r[0])


def test_philox_known_answers(golden):
    for v in golden["philox4x32_10"]:
        h = lambda a: [int(x, 16) if isinstance(x, str) else x for x in a]
        out = kkoracle.philox(h(v["ctr"]), h(v["key"]))
        assert [int(x) for x in out] == h(v["out"])


def test_sample_stream_mapping():
    """reference src/libcore/random.cpp:630-639: 23 mantissa bits -> [0,1); stream = philox(key=(pixel, seed.lo),
    counter=(sample, dim/4, seed.hi, 0))"""
    s = kkoracle.sample_stream(0x1234567800000005, 77, 9, 8)
    for blk in range(2):
        raw = kkoracle.philox([9, blk, 0x12345678, 0], [77, 5])
        exp = ((raw >> 9) | 0x3f800000).astype(np.uint32).view(np.float32) - 1.0
        np.testing.assert_array_equal(s[4 * blk:4 * blk + 4], exp)
    big = kkoracle.sample_stream(1, 3, 0, 4096)
    assert big.min() >= 0.0 and big.max() < 1.0 and abs(big.mean() - 0.5) < 0.02


@pytest.mark.parametrize("seed", [0, 1])
def test_kdtree_equals_brute_force(seed):
    """Havran traversal over the SAH kd-tree returns what a loop over all TriAccel records returns."""
    sc = helpers.triangle_soup(1500, seed=seed)
    o = kkoracle.Oracle(sc)
    rays = helpers.chord_rays(20000, (0, 0, 0), 8.0, seed=seed)
    a, b = o.trace(rays), o.trace(rays, brute=True)
    assert (a["prim"] != 0xFFFFFFFF).mean() > 0.2
    mism = a["prim"] != b["prim"]
    # a triangle hit exactly on a kd cell boundary may be found in either cell first; everything else is identical
    assert mism.sum() <= 2
    np.testing.assert_array_equal(a["t"][~mism], b["t"][~mism])
    sa, sb = o.trace(rays, any_hit=True), o.trace(rays, any_hit=True, brute=True)
    assert ((sa["prim"] != 0xFFFFFFFF) != (sb["prim"] != 0xFFFFFFFF)).sum() == 0


def test_adaptive_epsilon_rule():
    """reference src/librender/skdtree.cpp:126-132 vs :213-216"""
    sc = Scene(); m = sc.diffuse()
    sc.add_mesh([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], [[0, 1, 2]], m)
    sc.add_mesh([[-1, -1, 5], [1, -1, 5], [0, 1, 5]], [[0, 1, 2]], m)
    sc.set_sensor(np.eye(4), 45, 8, 8)
    o = kkoracle.Oracle(sc)
    # origin 1e-5 below the first triangle, |o| ~ 1e-5: closest-hit epsilon clamps to 1e-8 (hit), shadow epsilon = 1e-9 (hit)
    r = np.array([[0, 0, -1e-5, 1e-4, 0, 0, 1, np.inf]], np.float32)
    assert o.trace(r)["prim"][0] == 0
    # far from the origin the epsilon scales with |o|: mint = 1e-4 * 100 = 1e-2 skips a surface 1e-3 ahead
    sc2 = Scene(); m = sc2.diffuse()
    sc2.add_mesh([[99, -1, 0], [101, -1, 0], [100, 1, 0]], [[0, 1, 2]], m)
    sc2.add_mesh([[99, -1, 5], [101, -1, 5], [100, 1, 5]], [[0, 1, 2]], m)
    sc2.set_sensor(np.eye(4), 45, 8, 8)
    o2 = kkoracle.Oracle(sc2)
    r2 = np.array([[100, 0, -1e-3, 1e-4, 0, 0, 1, np.inf]], np.float32)
    assert o2.trace(r2)["prim"][0] == 1
    r2[0, 3] = 2e-4   # any other mint is taken literally
    assert o2.trace(r2)["prim"][0] == 0


# ---- microfacet distribution: reference src/tests/test_microfacet.cpp:57-72, 92-172 ----
MF_CASES = [(abi.DISTR_BECKMANN, 0.5, 0.5), (abi.DISTR_GGX, 0.5, 0.5), (abi.DISTR_PHONG, 0.5, 0.5),
            (abi.DISTR_BECKMANN, 0.5, 0.3), (abi.DISTR_GGX, 0.5, 0.3), (abi.DISTR_PHONG, 0.5, 0.3)]


def _chi2_sphere(samples_m, pdf_fn, n_theta=10, n_phi=20, fine=24):
    """chi^2 statistic of sampled unit vectors against a density over the upper hemisphere
    (bins as in reference src/tests/test_chisquare.cpp:399: 10 x 20)"""
    from scipy import stats
    ct = np.clip(samples_m[:, 2], -1, 1); ph = np.mod(np.arctan2(samples_m[:, 1], samples_m[:, 0]), 2 * np.pi)
    hist, _, _ = np.histogram2d(ct, ph, bins=[n_theta, n_phi], range=[[0, 1], [0, 2 * np.pi]])
    # expected: midpoint rule with fine x fine points per bin in (cos theta, phi), d omega = d(cos theta) d phi
    cs = (np.arange(n_theta * fine) + 0.5) / (n_theta * fine); ps = (np.arange(n_phi * fine) + 0.5) * 2 * np.pi / (n_phi * fine)
    C, P = np.meshgrid(cs, ps, indexing="ij")
    s = np.sqrt(1 - C ** 2)
    m = np.stack([s * np.cos(P), s * np.sin(P), C], -1).reshape(-1, 3)
    dens = pdf_fn(m).reshape(C.shape) * (1.0 / (n_theta * fine)) * (2 * np.pi / (n_phi * fine))
    exp = dens.reshape(n_theta, fine, n_phi, fine).sum((1, 3)) * len(samples_m)
    o, e = hist.ravel(), exp.ravel()
    order = np.argsort(e)
    o, e = o[order], e[order]
    # pool low-expectation cells (reference libcore/chisquare.cpp pools cells with expected frequency < 5)
    k = int(np.searchsorted(np.cumsum(e), 5.0)) + 1
    o = np.concatenate([[o[:k].sum()], o[k:]]); e = np.concatenate([[e[:k].sum()], e[k:]])
    chi2 = ((o - e) ** 2 / np.maximum(e, 1e-12)).sum()
    return stats.chi2.sf(chi2, len(o) - 1), float(exp.sum() / len(samples_m))


@pytest.mark.parametrize("distr,au,av", MF_CASES)
def test_microfacet_sample_all_matches_pdf(distr, au, av):
    rng = np.random.RandomState(3)
    u = rng.rand(200000, 2).astype(np.float32)
    wi = np.array([0, 0, 1], np.float32)
    s = kkoracle.microfacet_sample(distr, au, av, False, wi, u)
    m, pdf = s[:, :3], s[:, 3]
    assert np.abs(np.linalg.norm(m, axis=1) - 1).max() < 1e-4              # test_microfacet.cpp:66-68
    ref = kkoracle.microfacet_eval(distr, au, av, False, wi, m)[:, 1]
    ok = ref > 1e-6
    assert (np.abs(pdf[ok] - ref[ok]) / ref[ok]).max() < 2e-3            # :61-65 (1e-4 there in double builds; fp32 here)
    p, total = _chi2_sphere(m, lambda mm: kkoracle.microfacet_eval(distr, au, av, False, wi, mm)[:, 1])
    assert abs(total - 1) < 2e-2 and p > 0.0025 / 12                       # significance level of test_microfacet.cpp:27-39


@pytest.mark.parametrize("distr,au,av", [(abi.DISTR_BECKMANN, 0.3, 0.3), (abi.DISTR_GGX, 0.3, 0.3), (abi.DISTR_BECKMANN, 0.5, 0.3),
                                         (abi.DISTR_GGX, 0.5, 0.3), (abi.DISTR_BECKMANN, 0.1, 0.1), (abi.DISTR_GGX, 0.2, 0.3)])
def test_microfacet_sample_visible_matches_pdf(distr, au, av):
    rng = np.random.RandomState(5)
    for k in range(3):
        wi = rng.normal(size=3); wi[2] = abs(wi[2]) + 0.05; wi = (wi / np.linalg.norm(wi)).astype(np.float32)
        u = rng.rand(150000, 2).astype(np.float32)
        s = kkoracle.microfacet_sample(distr, au, av, True, wi, u)
        m = s[:, :3]
        assert np.abs(np.linalg.norm(m, axis=1) - 1).max() < 1e-4
        fine = 160 if min(au, av) < 0.15 else 24   # a peaked lobe needs a finer quadrature of the expected frequencies
        p, total = _chi2_sphere(m, lambda mm: kkoracle.microfacet_eval(distr, au, av, True, wi, mm)[:, 1], fine=fine)
        assert abs(total - 1) < 3e-2, total
        assert p > 0.0025 / 36, (distr, au, av, wi, p)


# ---- BSDF sample / eval / pdf consistency: reference src/tests/test_chisquare.cpp:163-203 (ERROR_REQ = 1e-2) ----
def _bsdf_scene():
    sc = Scene()
    ids = {
        "diffuse": sc.diffuse((0.5, 0.3, 0.2)),
        "roughconductor": sc.roughconductor((0.2, 0.92, 1.1), (3.9, 2.45, 2.14), alpha=0.3),
        "roughconductor_ggx_aniso": sc.roughconductor((0.2, 0.92, 1.1), (3.9, 2.45, 2.14), alpha=0.3, alpha_v=0.15, distribution=abi.DISTR_GGX),
        "roughconductor_all": sc.roughconductor((0.2, 0.92, 1.1), (3.9, 2.45, 2.14), alpha=0.3, sample_visible=False),
        "roughplastic": sc.roughplastic((0.4, 0.5, 0.3), alpha=0.2),
        "roughplastic_ggx_nl": sc.roughplastic((0.4, 0.5, 0.3), alpha=0.3, distribution=abi.DISTR_GGX, nonlinear=True),
    }
    ids["twosided"] = sc.twosided(ids["diffuse"], ids["roughplastic"])
    sc.add_mesh([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 1, 2]], ids["diffuse"])
    sc.set_sensor(np.eye(4), 45, 8, 8)
    return sc, ids


@pytest.fixture(scope="module")
def bsdf_oracle():
    sc, ids = _bsdf_scene()
    return kkoracle.Oracle(sc), ids


@pytest.mark.parametrize("name", ["diffuse", "roughconductor", "roughconductor_ggx_aniso", "roughconductor_all", "roughplastic",
                                  "roughplastic_ggx_nl", "twosided"])
def test_bsdf_sample_eval_pdf_agree(bsdf_oracle, name):
    o, ids = bsdf_oracle
    mat = ids[name]
    rng = np.random.RandomState(11)
    for wi in ([0.3, -0.2, 0.93], [0.7, 0.1, 0.4], [-0.2, 0.5, -0.84] if name == "twosided" else [0.0, 0.0, 1.0]):
        wi = np.float32(wi) / np.linalg.norm(wi)
        u = rng.rand(4000, 2).astype(np.float32)
        s = o.bsdf_sample(mat, wi, u)
        checked = 0
        for row in s[:400]:
            wo, w, pdf = row[0:3], row[3:6], row[6]
            if not w.any() or pdf == 0:
                continue
            f, p = o.bsdf_eval(mat, wi, wo)
            assert abs(p - pdf) <= 1e-2 * max(pdf, 1e-3), (name, wi, wo, p, pdf)
            np.testing.assert_allclose(f / p, w, rtol=1e-2, atol=1e-4)
            checked += 1
        assert checked > 100
        # energy: E[weight] <= 1 per channel (plus Monte Carlo slack)
        assert s[:, 3:6].mean(0).max() < 1.02


def test_dielectric_delta_lobes(bsdf_oracle):
    """reference src/bsdfs/dielectric.cpp:277-308: reflect with probability F, else refract with (1/eta)^2 scaling"""
    sc = Scene(); d = sc.dielectric(1.5, 1.0)
    sc.add_mesh([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 1, 2]], d); sc.set_sensor(np.eye(4), 45, 8, 8)
    o = kkoracle.Oracle(sc)
    wi = np.float32([0.6, 0, 0.8])
    u = np.random.RandomState(0).rand(20000, 2).astype(np.float32)
    s = o.bsdf_sample(d, wi, u)
    refl = s[:, 7] == 0x40
    F = refl.mean()
    cos_t = np.sqrt(1 - (0.6 / 1.5) ** 2)
    rs = (0.8 - 1.5 * cos_t) / (0.8 + 1.5 * cos_t); rp = (1.5 * 0.8 - cos_t) / (1.5 * 0.8 + cos_t)
    assert abs(F - 0.5 * (rs * rs + rp * rp)) < 0.01
    np.testing.assert_allclose(s[refl][0, 0:3], [-0.6, 0, 0.8], atol=1e-6)
    t = s[~refl][0]
    np.testing.assert_allclose(t[0:3], [-0.4, 0, -cos_t], atol=1e-5)
    np.testing.assert_allclose(t[3:6], [1 / 2.25] * 3, rtol=1e-5)
    f, p = o.bsdf_eval(d, wi, t[0:3])
    assert not f.any() and p == 0    # nothing in the solid-angle measure (dielectric.cpp:228-253)


def test_rough_transmittance_properties():
    """reference src/tests/test_rtrans.cpp:154-190 accepts 1e-3 absolute error against brute-force integration of the
    rough dielectric; here the table is that integration.  Check limits: alpha -> 0 gives the Fresnel transmittance,
    values lie in [0,1] and grow with cos(theta)."""
    tab, diff = kkoracle.rough_transmittance(abi.DISTR_BECKMANN, 0.1, 1.49)
    assert tab.min() >= 0 and tab.max() <= 1 and np.all(np.diff(tab[10:]) > -1e-3)
    for mu in (0.3, 0.6, 0.9):
        t = kkoracle.rough_transmittance_at(abi.DISTR_GGX, 1e-3, 1.5, mu)
        ct = np.sqrt(1 - (1 - mu * mu) / 2.25)
        rs = (mu - 1.5 * ct) / (mu + 1.5 * ct); rp = (1.5 * mu - ct) / (1.5 * mu + ct)
        assert abs(t - (1 - 0.5 * (rs * rs + rp * rp))) < 2e-3
    assert 0.8 < diff < 1.0


def test_rfilter_table_gaussian():
    """reference src/libcore/rfilter.cpp:37-55 + src/rfilters/gaussian.cpp:32-57"""
    v, scale, border = kkoracle.rfilter_table(abi.RFILTER_GAUSSIAN, 2.0, (0.5, 0.0))
    assert border == 2 and abs(scale - 15.5) < 1e-6 and v[31] == 0
    assert abs(2 * v[:31].sum() * (2.0 / 31) - 1) < 1e-5        # normalised to unit integral
    assert np.all(np.diff(v[:31]) < 0)
    v, scale, border = kkoracle.rfilter_table(abi.RFILTER_BOX, 0.5 + 1e-5)
    assert border == 1 and np.allclose(v[:31], v[0])


def test_emitter_sampling_matches_pdf():
    """Emitter::sampleDirect vs pdfDirect (reference src/tests/test_chisquare.cpp:575-619 idea): the Monte Carlo
    estimate of int Le dw equals the analytic form-factor integral for a quad seen from a point."""
    sc = Scene(); blk = sc.diffuse((0, 0, 0))
    q = np.array([[-1, 2, -1], [-1, 2, 1], [1, 2, 1], [1, 2, -1]], np.float32)
    sc.add_mesh(q, [[0, 2, 1], [0, 3, 2]], blk, radiance=(3, 3, 3))
    sc.add_point_light((0, -5, 0), (8, 8, 8))
    sc.set_sensor(np.eye(4), 45, 8, 8)
    o = kkoracle.Oracle(sc)
    u = np.random.RandomState(2).rand(200000, 2).astype(np.float32)
    s = o.sample_emitter_direct([0, 0, 0], [0, 1, 0], u)
    area = s[s[:, 8] == 0]
    # value = Le / pdf (incl. 1/2 selection probability): E[value * [emitter==area]] = Le * solid angle of the quad
    est = s[:, 0] * (s[:, 8] == 0)
    a = 1.0; h = 2.0   # solid angle of a 2a x 2a square at height h: 4 asin(a^2 / (a^2 + h^2))
    omega = 4 * np.arcsin(a * a / (a * a + h * h))
    assert abs(est.mean() - 3 * omega) < 0.02 * 3 * omega
    pt = s[s[:, 8] == 1]
    np.testing.assert_allclose(pt[:, 0], 8 / 25 / 0.5, rtol=1e-5)   # intensity / dist^2 / selection pdf
    assert abs(len(area) / len(s) - 0.5) < 0.01


def test_envmap_sampling_matches_pdf():
    """reference src/emitters/envmap.cpp:567-633: internalSampleDirection's pdf equals internalPdfDirection, and the
    importance-sampled estimate of the env integral equals the quadrature of the texels."""
    rng = np.random.RandomState(4)
    tex = (rng.rand(16, 32, 3) ** 4 * 5).astype(np.float32); tex[3, 7] = 200
    sc = Scene(); m = sc.diffuse()
    for x in (-10.0, 10.0):   # two specks far apart: a bounding sphere of radius ~15 and practically no occlusion
        sc.add_mesh([[x, 0, 0], [x + 1e-3, 0, 0], [x, 1e-3, 0]], [[0, 1, 2]], m)
    sc.set_sensor(np.eye(4), 45, 8, 8)
    sc.set_envmap(tex)
    o = kkoracle.Oracle(sc)
    u = rng.rand(100000, 2).astype(np.float32)
    s = o.sample_emitter_direct([0, 0, 0], [0, 0, 0], u)
    ok = s[:, 7] > 0
    assert ok.mean() > 0.99
    probe = o.env_probe(s[ok, 3:6])
    # away from the poles (where 1 - d.y^2 cancels in single precision, in the reference as well) both agree
    mid = np.abs(s[ok, 4]) < 0.995
    np.testing.assert_allclose(probe[mid, 3], s[ok, 7][mid], rtol=5e-3)
    np.testing.assert_allclose(probe[mid, 0:3], (s[ok, 0:3] * s[ok, 7:8])[mid], rtol=5e-3, atol=1e-4)
    # E[Le/pdf] = int Le dw: compare with a midpoint quadrature of evalEnvironment
    th = (np.arange(256) + 0.5) * np.pi / 256; ph = (np.arange(512) + 0.5) * 2 * np.pi / 512
    T, P = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([np.sin(P) * np.sin(T), np.cos(T), -np.cos(P) * np.sin(T)], -1).reshape(-1, 3)
    val = o.env_probe(d)[:, 0].reshape(T.shape)
    integral = (val * np.sin(T)).sum() * (np.pi / 256) * (2 * np.pi / 512)
    assert abs(s[:, 0].mean() - integral) < 0.03 * integral


def test_render_tile_and_sample_sharding_sum_to_full():
    """renderBlock semantics (reference src/librender/integrator.cpp:140-188, renderproc.cpp:142-149): the film is the
    sum of tile blocks including borders, and of disjoint sample ranges."""
    sc = scenes.cornell_box(48, 40)
    o = kkoracle.Oracle(sc)
    full, st = o.render(4, seed=3, threads=2)
    assert st.samples == 48 * 40 * 4
    parts = [o.render(4, seed=3, threads=2, tile_index=k, tile_count=3, block_size=16)[0] for k in range(3)]
    np.testing.assert_allclose(sum(parts), full, rtol=1e-5, atol=1e-6)
    halves = [o.render(2, seed=3, threads=2, sample_first=k * 2)[0] for k in range(2)]
    np.testing.assert_allclose(sum(halves), full, rtol=1e-5, atol=1e-6)
    # film weight: every interior pixel receives total filter weight ~ spp
    w = full[8:-8, 8:-8, 4]
    assert abs(w.mean() - 4) < 0.2


def test_white_furnace_diffuse_plane_under_constant_env():
    """Analytic check (hard parts iv): a diffuse plane with albedo rho under a constant environment of
    radiance 1 reflects exactly rho towards the camera, whatever the MIS weights do."""
    sc = Scene()
    m = sc.diffuse((0.6, 0.4, 0.2))
    sc.add_mesh([[-50, 0, -50], [-50, 0, 50], [50, 0, 50], [50, 0, -50]], [[0, 1, 2], [0, 2, 3]], m)
    sc.set_envmap(np.ones((8, 16, 3), np.float32))
    from kkdirect import look_at
    sc.set_sensor(look_at((0, 3, 0), (0, 0, 0.01), (0, 0, 1)), 30, 16, 16, rfilter=abi.RFILTER_BOX)
    film, _ = kkoracle.Oracle(sc).render(256, seed=5, threads=2)
    img = helpers.develop(film)
    np.testing.assert_allclose(img.mean((0, 1)), [0.6, 0.4, 0.2], rtol=0.02)
