"""The entry point that allocates nothing, kkd_trace_device_noalloc (include/kkdirect.h): called through the C ABI and
compared with kkd_trace_device -- hits bit for bit, ray statistics equal -- on two calls in a row, the second with a
smaller input (the library's static scratch array keeps the first call's counters, so the second call shows that they are
cleared).  Device buffers are the caller's (torch tensors); the call itself must not allocate."""
import ctypes as C
import numpy as np
import pytest
import torch   # before libkkdirect is loaded: torch brings its own copy of the HIP runtime, and the one loaded first serves both
from kkdirect import abi, scenes
import helpers

pytestmark = pytest.mark.gpu


def _rays(sc, n, seed):
    lo, hi = sc.aabb()
    return helpers.chord_rays(n, (lo + hi) / 2, float(np.linalg.norm(hi - lo)) / 2, seed=seed)


@pytest.mark.parametrize("any_hit", [False, True])
def test_noalloc_trace_equals_trace_device_twice(gpu, any_hit):
    sc = scenes.sponza_class(64, 36, target_tris=20000)
    with gpu.Context(device=0).upload(sc) as ctx:
        for n, seed in ((200_000, 1), (3_001, 2)):              # second call: smaller input, same scratch array
            rays = torch.from_numpy(_rays(sc, n, seed)).cuda()
            out_a = torch.zeros(n * 4, dtype=torch.int32, device="cuda")
            out_b = torch.full((n * 4,), -7, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            sa = ctx.trace_device(rays.data_ptr(), n, out_a.data_ptr(), any_hit=any_hit)
            free_before = torch.cuda.mem_get_info()[0]            # hipMemGetInfo: device-wide, whoever allocates
            sb = ctx.trace_device_noalloc(rays.data_ptr(), n, out_b.data_ptr(), any_hit=any_hit)
            assert torch.cuda.mem_get_info()[0] >= free_before
            a = out_a.cpu().numpy().view(gpu.HIT_DTYPE); b = out_b.cpu().numpy().view(gpu.HIT_DTYPE)
            if any_hit:                                           # only hit / miss is defined for occlusion queries
                assert np.array_equal(a["prim"] != 0xFFFFFFFF, b["prim"] != 0xFFFFFFFF)
            else:
                assert a.tobytes() == b.tobytes()
            assert (a["prim"] != 0xFFFFFFFF).sum() > n // 10
            cls = abi.RAYS_SHADOW if any_hit else abi.RAYS_BSDF
            assert sb.class_rays[cls] == sa.class_rays[cls] == n
            assert (sb.rays_closest, sb.rays_shadow) == (sa.rays_closest, sa.rays_shadow)


def test_noalloc_rejects_what_it_cannot_take(gpu):
    sc = scenes.cornell_box(16, 16)
    with gpu.Context(device=0).upload(sc) as ctx:
        st = abi.Stats()
        rc = ctx.lib.kkd_trace_device_noalloc(ctx.h, C.c_void_p(16), abi.NOALLOC_MAX_RAYS + 1, 0, C.c_void_p(16), C.byref(st))
        assert rc == abi.KKD_ERR_UNSUPPORTED and b"KKD_NOALLOC_MAX_RAYS" in ctx.lib.kkd_last_error(ctx.h)
        assert ctx.lib.kkd_trace_device_noalloc(ctx.h, None, 5, 0, None, C.byref(st)) == abi.KKD_ERR_INVALID
        assert ctx.lib.kkd_trace_device_noalloc(ctx.h, None, 0, 0, None, C.byref(st)) == abi.KKD_OK and st.class_rays[abi.RAYS_BSDF] == 0
    with gpu.Context(device=0) as empty:
        assert empty.lib.kkd_trace_device_noalloc(empty.h, C.c_void_p(16), 1, 0, C.c_void_p(16), None) == abi.KKD_ERR_INVALID   # no scene
