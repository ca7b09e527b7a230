"""KKD_FLAG_GPU_BUILD (run with -m gpu): the acceleration structure built on the device (Morton-ordered linear BVH,
kkd_gpu_build.hip) must answer every ray like the host-built binned-SAH tree -- the hit does not depend on the tree, only
which of several triangles at exactly the same distance is reported can (the walk order differs) -- and a render through it
must produce the same film.  C2 and C4 at full size as well, with the build times recorded."""
import time
import numpy as np
import pytest
import torch   # before libkkdirect is loaded (see test_zz_no_alloc.py)
import kkoracle
from kkdirect import Scene, abi, scenes
import helpers
from test_gpu_parity import _material_scene

pytestmark = pytest.mark.gpu
MISS = 0xFFFFFFFF


def _rays(sc, n, seed):
    lo, hi = sc.aabb()
    chords = helpers.chord_rays(n, (lo + hi) / 2, float(np.linalg.norm(hi - lo)) / 2, seed=seed)
    rng = np.random.RandomState(seed + 1)
    inside = np.zeros((n, 8), np.float32)
    inside[:, 0:3] = lo + (hi - lo) * rng.rand(n, 3); inside[:, 3] = 1e-4
    d = rng.normal(size=(n, 3)); inside[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True); inside[:, 7] = np.inf
    return np.concatenate([chords, inside])


def _same_answers(gpu, sc, rays, **kw):
    with gpu.Context(device=0, **kw).upload(sc) as host, gpu.Context(device=0, gpu_build=True, **kw).upload(sc) as dev:
        assert dev.accel_info()["n_leaf_tris"] == host.accel_info()["n_leaf_tris"] == sc._nt
        a, b = host.trace(rays), dev.trace(rays)
        assert np.array_equal(a["prim"] == MISS, b["prim"] == MISS)
        assert np.array_equal(a["t"].view(np.uint32), b["t"].view(np.uint32))          # the same triangle arithmetic: the same distance, bit for bit
        differ = a["prim"] != b["prim"]                                               # only equal-distance ties may name another triangle
        assert differ.mean() <= 1e-4, differ.mean()
        n_bad, unexplained, _ = helpers.explain_mismatches(sc, rays, a, b)
        assert unexplained == 0
        sa, sb = host.trace(rays, any_hit=True), dev.trace(rays, any_hit=True)
        assert np.array_equal(sa["prim"] == MISS, sb["prim"] == MISS)
        return host.accel_info(), dev.accel_info()


@pytest.mark.parametrize("name", ["cornell", "soup", "sponza", "materials"])
def test_device_built_tree_answers_like_the_host_built_one(gpu, name):
    sc = {"cornell": lambda: scenes.cornell_box(64, 64), "soup": lambda: helpers.triangle_soup(6000, seed=3, normals=True),
          "sponza": lambda: scenes.sponza_class(96, 54, target_tris=60_000), "materials": lambda: _material_scene(64, 48)}[name]()
    _same_answers(gpu, sc, _rays(sc, 200_000, seed=5))
    with gpu.Context(device=0, seed=3).upload(sc) as host, gpu.Context(device=0, seed=3, gpu_build=True).upload(sc) as dev:
        fa, sa = host.render(8); fb, sb = dev.render(8)
    assert sa.samples == sb.samples and sa.rays_closest == sb.rays_closest
    np.testing.assert_allclose(fb, fa, rtol=1e-4, atol=1e-5)


def test_device_build_edge_cases(gpu):
    """one triangle, two triangles, thousands of coincident triangles (equal Morton codes: the hierarchy tells them apart by
    position, and the tree must stay within the traversal stack), a degenerate triangle, projection-test mode"""
    def tri_scene(tris):
        sc = Scene(); pos = np.asarray(tris, np.float32).reshape(-1, 3)
        sc.add_mesh(pos, np.arange(len(pos), dtype=np.uint32).reshape(-1, 3), sc.diffuse())
        sc.set_sensor(np.eye(4), 45, 8, 8)
        return sc
    one = [[-1, 0, -1], [1, 0, -1], [0, 0, 1.5]]
    rays = np.array([[0, 1, 0, 0, 0, -1, 0, np.inf], [5, 1, 0, 0, 0, -1, 0, np.inf], [0.2, -1, 0.1, 0, 0, 1, 0, np.inf]], np.float32)
    for tris in (one, one + [[2, 0, 2], [3, 0, 2], [2, 0, 3]], one * 5000, one + [[4, 4, 4], [4, 4, 4], [5, 5, 5]]):
        sc = tri_scene(tris)
        with gpu.Context(device=0, gpu_build=True).upload(sc) as dev:
            h = dev.trace(rays)
            assert h["prim"][0] != MISS and h["prim"][1] == MISS and h["prim"][2] != MISS
            np.testing.assert_allclose(h["t"][[0, 2]], 1.0, rtol=1e-6)
            film, st = dev.render(2)
            assert st.samples == 8 * 8 * 2 and np.isfinite(film).all()
    sc = scenes.cornell_box(128, 128)
    o = kkoracle.Oracle(sc)
    cam = o.camera_rays(np.random.RandomState(1).rand(200_000, 2).astype(np.float32) * 128)
    with gpu.Context(device=0, gpu_build=True, projection_test=True).upload(sc) as dev:
        a, b = o.trace(cam), dev.trace(cam)
    assert (a["prim"] != b["prim"]).sum() <= 2
    same = a["prim"] == b["prim"]
    for f in ("t", "u", "v"):
        assert np.array_equal(a[f][same].view(np.uint32), b[f][same].view(np.uint32))
    bad = scenes.cornell_box(16, 16); bad._pos[0][0, 0] = np.nan
    with pytest.raises(gpu.KKDError, match="non-finite"):
        gpu.Context(device=0, gpu_build=True).upload(bad)


@pytest.mark.parametrize("workload", ["c2", "c4"])
def test_device_build_at_full_size(gpu, workload, record_property):
    sc = scenes.sponza_class(1920, 1080) if workload == "c2" else scenes.scanned_mesh(3840, 2160)
    o_rays = _rays(sc, 150_000, seed=9)
    t0 = time.perf_counter(); host = gpu.Context(device=0).upload(sc); t_host = time.perf_counter() - t0
    t0 = time.perf_counter(); dev = gpu.Context(device=0, gpu_build=True).upload(sc); t_dev = time.perf_counter() - t0
    try:
        a, b = host.trace(o_rays), dev.trace(o_rays)
        assert np.array_equal(a["t"].view(np.uint32), b["t"].view(np.uint32))
        assert (a["prim"] != b["prim"]).mean() <= 1e-4 and helpers.explain_mismatches(sc, o_rays, a, b)[1] == 0
        _, sa = host.trace(o_rays, counted=True); _, sb = dev.trace(o_rays, counted=True)
        ia, ib = host.accel_info(), dev.accel_info()
        report = "%s: upload host-built %.2f s, device-built %.2f s; nodes %d vs %d; nodes visited per ray %.1f vs %.1f" % (
            workload, t_host, t_dev, ia["n_nodes"], ib["n_nodes"], sa.nodes_visited / len(o_rays), sb.nodes_visited / len(o_rays))
        print(report); record_property("build", report)
        assert t_dev < t_host
        t = torch.zeros(dev.width * dev.height * 5, dtype=torch.float32, device="cuda")
        st = dev.render_device(t.data_ptr(), 4)
        assert st.samples == dev.width * dev.height * 4 and bool(torch.isfinite(t).all())
    finally:
        host.close(); dev.close()
