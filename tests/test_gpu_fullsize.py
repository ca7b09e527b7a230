"""Full-size GPU tests of the scale-defining configurations of BASELINE.json (run with -m gpu on an MI355X):
C2 (255 k triangles, 1080p), C3 (C2 geometry + 10 000 single-triangle emitters: a 10 001-entry emitter CDF) and C4
(10.7 M triangles, 4096x2048 environment map, 4K) at the sizes the configs name -- 32-bit offset arithmetic, BVH depth
against the traversal stack, CDF / guide-table indexing at full resolution are all under an assertion here.

  * hit ids on fixed ray sets against the oracle's ShapeKDTree restatement built on the FULL scene, both leaf tests
    (north-star bar: bit-exact except documented ties, <= 1e-5 of the rays); every differing ray is re-evaluated in float64
    and must be a tie there (helpers.explain_mismatches);
  * same-sample-stream image parity with the oracle on a crop window of the full-resolution film (the oracle cannot
    render 4K x 512 spp in test time; the crop renders the same pixels of the same film with the same sample stream);
  * size-independent properties of the full frame: exact sample / ray counts, film weight, finiteness, tile-sharded and
    sample-split partial films summing to the whole.
"""
import numpy as np
import pytest
import kkoracle
from kkdirect import abi, scenes
import helpers

pytestmark = pytest.mark.gpu
MISS = 0xFFFFFFFF


def _ray_sets(o, sc, n, seed, chord_n=0):
    rng = np.random.RandomState(seed)
    cam = o.camera_rays(rng.rand(n, 2).astype(np.float32) * [o.width, o.height])
    h = o.trace(cam)
    ok = h["prim"] != MISS
    p = cam[ok, 0:3] + cam[ok, 4:7] * h["t"][ok, None]
    d = rng.normal(size=(len(p), 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    sec = np.zeros((len(p), 8), np.float32)
    sec[:, 0:3] = p; sec[:, 3] = 1e-4; sec[:, 4:7] = d; sec[:, 7] = np.inf      # mint = Epsilon: the adaptive-epsilon case
    sets = {"camera": (cam, h), "surface-leaving": (sec, None)}
    if chord_n:
        lo, hi = sc.aabb()
        sets["chords"] = (helpers.chord_rays(chord_n, (lo + hi) / 2, float(np.linalg.norm(hi - lo)) / 2, seed=seed + 1), None)
    return sets


def _hit_id_bar(gpu, sc, o, sets, report, watertight_bounds=(5e-5, 5e-4)):
    """projection mode: <= 1e-5 differing ids, every one a float64 tie, (t, u, v) bit-equal elsewhere.  watertight mode
    (the default): zero unexplained differences; the rate is recorded and bounded (camera rays, other rays) by what the
    reference's own single-precision projection test resolves on this scene's triangles -- DESIGN.md section 4 states
    that the default leaf test does not meet the 1e-5 bar on dense scenes and why."""
    for proj in (True, False):
        with gpu.Context(device=0, projection_test=proj).upload(sc) as ctx:
            for name, (rays, ref) in sets.items():
                a = ref if ref is not None else o.trace(rays)
                b = ctx.trace(rays)
                n_bad, unexplained, classes = helpers.explain_mismatches(sc, rays, a, b)
                rate = n_bad / len(rays)
                report.append("%s %s rays=%d differing=%d rate=%.2e %s" % ("projection" if proj else "watertight", name, len(rays), n_bad, rate, classes))
                assert unexplained == 0, report[-1]
                same = a["prim"] == b["prim"]
                if proj:
                    assert n_bad <= max(1, int(1e-5 * len(rays))), report[-1]
                    for f in ("t", "u", "v"):
                        assert np.array_equal(a[f][same].view(np.uint32), b[f][same].view(np.uint32)), (name, f)
                else:
                    assert rate <= (watertight_bounds[0] if name == "camera" else watertight_bounds[1]), report[-1]
                    hit = same & (a["prim"] != MISS)
                    far = np.abs(b["t"][hit] - a["t"][hit]) > 1e-3 * np.abs(a["t"][hit]) + 1e-4
                    assert far.mean() < 5e-3, (name, far.mean())
                sa, sb = o.trace(rays, any_hit=True), ctx.trace(rays, any_hit=True)
                occ = int(((sa["prim"] != MISS) != (sb["prim"] != MISS)).sum())
                assert occ <= max(1, int((1e-5 if proj else watertight_bounds[1]) * len(rays))), (name, occ)


def _crop_parity(gpu, sc, crop, spp, tol, **kw):
    """the same pixels of the full-resolution film, same sample stream, GPU vs oracle"""
    s = sc.sensor
    s.crop_offset_x, s.crop_offset_y, s.crop_width, s.crop_height = crop
    try:
        ref, so = kkoracle.Oracle(sc).render(spp, seed=1, **kw)
        with gpu.Context(device=0, seed=1).upload(sc) as ctx:
            film, sg = ctx.render(spp, **kw)
    finally:
        s.crop_offset_x, s.crop_offset_y, s.crop_width, s.crop_height = 0, 0, s.film_width, s.film_height
    assert sg.samples == so.samples == crop[2] * crop[3] * spp
    assert np.isfinite(film).all() and film.min() >= 0
    r = helpers.rel_mse(helpers.develop(film), helpers.develop(ref))
    helpers.record_parity(r, tol)
    assert r < tol, r
    return r


def _frame_properties(ctx, W, H, spp, tiles=(3,), split=True):
    n = W * H * spp
    film, st = ctx.render(spp)
    assert st.samples == n and st.class_rays[abi.RAYS_PRIMARY] == n
    assert st.class_rays[abi.RAYS_BSDF] <= n and st.class_rays[abi.RAYS_SHADOW] <= n and st.class_rays[abi.RAYS_BSDF] > n // 10
    assert np.isfinite(film).all() and film.min() >= 0
    assert abs(film[..., 4].sum() / n - 1) < 0.01
    assert abs(film[4:-4, 4:-4, 4].mean() - spp) < 0.01 * spp + 0.5
    for k in tiles:                              # tile-sharded partial films (with their filter borders) sum to the whole
        parts = [ctx.render(spp, tile_index=r, tile_count=k) for r in range(k)]
        assert sum(p[1].samples for p in parts) == n
        np.testing.assert_allclose(sum(p[0] for p in parts), film, rtol=2e-3, atol=2e-3)
    if split:                                    # sample-split partial films sum to the whole
        a, _ = ctx.render(spp // 2, sample_first=0); b, _ = ctx.render(spp - spp // 2, sample_first=spp // 2)
        np.testing.assert_allclose(a + b, film, rtol=2e-3, atol=2e-3)
    return film, st


def test_full_size_c2_hit_id_bar_both_modes(gpu, record_property):
    sc = scenes.sponza_class(1920, 1080)
    o = kkoracle.Oracle(sc)
    report = []
    _hit_id_bar(gpu, sc, o, _ray_sets(o, sc, 2_000_000, seed=0, chord_n=1_000_000), report)
    record_property("hit_id_report", report)
    print("\\n".join(report))


def test_full_size_c3_many_lights(gpu, record_property):
    """C3 at its full size: C2's geometry + 10 000 single-triangle area emitters (Scene::m_emitterPDF with 10 001 entries,
    librender/scene.cpp:376-380), 1920x1080, 128 spp."""
    sc = scenes.many_lights(1920, 1080, interior_tris=260_000)
    assert len(sc.emitters) >= 10_000
    o = kkoracle.Oracle(sc)
    report = []
    _hit_id_bar(gpu, sc, o, _ray_sets(o, sc, 300_000, seed=2), report)
    print("\\n".join(report))
    # measured 6.2e-11 with no flipped sample (profiles/parity_r3.json) and 8e-7 on the single-mesh variant below (profiles/
    # parity_same_stream_r3.json; one sample that takes another branch on a last-bit difference costs ~1e-7 to 1e-6): 12x measured
    r = _crop_parity(gpu, sc, (832, 476, 256, 128), 16, 1e-5)
    record_property("crop_relmse", r)
    with gpu.Context(device=0, seed=1).upload(sc) as ctx:
        _frame_properties(ctx, 1920, 1080, 128)
    # the variant with ONE emissive mesh of 10 000 triangles (TriMesh area CDF, librender/trimesh.cpp:392-407)
    sc1 = scenes.many_lights(1920, 1080, interior_tris=260_000, single_mesh=True)
    _crop_parity(gpu, sc1, (832, 476, 256, 128), 16, 1e-5)


_C4 = {}


def _c4_scene():
    """C4 / C5 share one scene: generated once per session (10.7 M triangles take a while in numpy)"""
    if "sc" not in _C4:
        _C4["sc"] = scenes.scanned_mesh(3840, 2160)
    return _C4["sc"]


def test_full_size_c4_scanned_mesh(gpu, record_property):
    """C4 at its full size: 10.7 M triangles (roughconductor GGX mesh, dielectric objects, ground) under a 4096x2048 HDR
    environment map with the two-level sampling table, 3840x2160; the frame properties at a reduced spp (the full 512 spp
    frame is what bench.py times)."""
    sc = _c4_scene()
    assert sc._nt > 10_000_000
    o = kkoracle.Oracle(sc)
    report = []
    # sub-pixel triangles far from the origin: the share of rays within fp32 resolution of an edge is larger than on C2
    _hit_id_bar(gpu, sc, o, _ray_sets(o, sc, 300_000, seed=3, chord_n=200_000), report, watertight_bounds=(4e-4, 2e-3))
    print("\\n".join(report))
    record_property("hit_id_report", report)
    # measured 6.7e-7, ALL of it from 273 of 524 288 samples (5e-4) that take another branch on a last-bit difference -- a hit
    # across an edge of the sub-pixel triangles, the lobe a sample picks, a Newton step -- and are then different, equally valid
    # samples under the 1e4 suns; without them 2.7e-11 (profiles/parity_r3.json classifies the 20 worst pixels): 15x measured
    r = _crop_parity(gpu, sc, (1792, 1016, 256, 128), 16, 1e-5)
    record_property("crop_relmse", r)
    with gpu.Context(device=0, seed=1).upload(sc) as ctx:
        info = ctx.accel_info()
        assert info["n_leaf_tris"] == sc._nt
        _frame_properties(ctx, 3840, 2160, 32, tiles=(8,))


def test_c5_partitioning_at_4096_spp(gpu, record_property):
    """C5 = the C4 scene at 4096 spp, tile-sharded vs sample-split (BASELINE.json configs[4]).  The test box has one GPU, so
    device 0 is listed EIGHT times: eight contexts with their own streams, scene copies and films render their shares side by
    side, and the merge code of both modes (tile gather with borders, film add; the reference's counterpart is
    BlockedRenderProcess::processResult -> Film::put, librender/renderproc.cpp:142-149, over the work units of
    librender/imageproc.cpp:28-78) runs for real, at the config's own sample count on a crop window of the 4K film.  What
    this cannot show is the peer copy between distinct devices and RCCL with more than one rank: the scaling curve on
    hardware stays the driver's to measure."""
    sc = _c4_scene()
    s = sc.sensor
    crop = (1792, 1016, 256, 128)
    s.crop_offset_x, s.crop_offset_y, s.crop_width, s.crop_height = crop
    spp, n = 4096, crop[2] * crop[3] * 4096
    try:
        with gpu.Context(device=0, seed=9).upload(sc) as ctx:
            full, st = ctx.render(spp)
        assert st.samples == n and np.isfinite(full).all() and full.min() >= 0
        films = {}
        for split in ("tile", "sample"):
            with gpu.MultiContext([0] * 8, split=split, seed=9).upload(sc) as m:
                films[split], sm = m.render(spp)
                assert m.merge_path == ("tile-gather" if split == "tile" else "copy-add"), m.merge_path
                assert "listed more than once" in m.diagnostics, m.diagnostics
                record_property("diagnostics_" + split, m.diagnostics)
            assert sm.samples == n and sm.class_rays[abi.RAYS_PRIMARY] == n
            assert sm.class_rays[abi.RAYS_BSDF] == st.class_rays[abi.RAYS_BSDF] and sm.class_rays[abi.RAYS_SHADOW] == st.class_rays[abi.RAYS_SHADOW]
            # same samples, other order of the float additions into a pixel (4096 samples x 25 pixels of footprint each)
            np.testing.assert_allclose(films[split], full, rtol=2e-4, atol=2e-3)
        np.testing.assert_allclose(films["tile"], films["sample"], rtol=2e-4, atol=2e-3)
        assert abs(full[4:-4, 4:-4, 4].mean() - spp) < 0.01 * spp
    finally:
        s.crop_offset_x, s.crop_offset_y, s.crop_width, s.crop_height = 0, 0, s.film_width, s.film_height
