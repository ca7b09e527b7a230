"""Ad-hoc first GPU run: hit-ID parity + Cornell image parity vs the oracle."""
import sys, os, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'oracle')); sys.path.insert(0, os.path.join(R, 'ray-tracer_amd'))
import kkoracle, kkdirect
from kkdirect import scenes
print('devices', kkdirect.device_count(), flush=True)
sc = scenes.cornell_box(256, 256)
o = kkoracle.Oracle(sc)
ctx = kkdirect.Context(seed=1).upload(sc)
print('accel', ctx.accel_info(), flush=True)
rng = np.random.RandomState(0)
sp = rng.rand(200000, 2).astype(np.float32) * 256
rays = o.camera_rays(sp)
t = time.time(); a = o.trace(rays); print('oracle trace s', time.time() - t)
b, st = ctx.trace(rays, counted=True)
print('gpu trace: nodes/ray', st.nodes_visited / len(rays), 'tris/ray', st.tris_tested / len(rays), 'sec', st.seconds)
mis = a['prim'] != b['prim']
print('prim mismatches', mis.sum(), 'of', len(rays), 'max |dt|', np.abs(a['t'][~mis] - b['t'][~mis])[np.isfinite(a['t'][~mis])].max())
if mis.sum():
    idx = np.nonzero(mis)[0][:10]
    for i in idx: print(i, a[i], b[i])
# any-hit
sh = o.trace(rays, any_hit=True); sg = ctx.trace(rays, any_hit=True)
print('anyhit mismatches', ((sh['prim'] != 0xFFFFFFFF) != (sg['prim'] != 0xFFFFFFFF)).sum())
# image
t = time.time(); fo, so = o.render(64, seed=1); print('oracle render s', time.time() - t, so.samples, so.rays_closest, so.rays_shadow)
t = time.time(); fg, sg = ctx.render(64); print('gpu render s', time.time() - t, 'dev s', sg.seconds, sg.samples, sg.rays_closest, sg.rays_shadow, 'trace s', sg.seconds_trace)
io, ig = kkdirect.develop(fo), kkdirect.develop(fg)
print('mean oracle', io.mean((0, 1)), 'gpu', ig.mean((0, 1)))
print('weight sum', fo[..., 4].sum(), fg[..., 4].sum())
rel = ((ig - io) ** 2 / (io ** 2 + 1e-3)).mean()
print('relMSE (same seed, both noisy)', rel, 'max abs', np.abs(ig - io).max())
np.save(os.path.join(R, 'gpurun_out', 'cbox_gpu.npy'), ig); np.save(os.path.join(R, 'gpurun_out', 'cbox_ora.npy'), io)
# per-sample Li parity
pix = np.stack([rng.randint(0, 256, 20000), rng.randint(0, 256, 20000), rng.randint(0, 64, 20000)], 1).astype(np.uint32)
lo = o.li_samples(pix, seed=1)
print('li oracle mean', lo[:, :3].mean(0))
