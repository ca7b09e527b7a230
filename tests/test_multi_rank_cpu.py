"""world_size-2 test of the multi-GPU decomposition on CPU (gloo): the film of the full job equals the sum-reduce of
the per-rank films, in tile-sharded mode (tile_index = rank, tile_count = world) and in sample-split mode
(sample_first = rank * spp).  The renderer here is the oracle, whose work decomposition parameters are the same
kkd_render_params the GPU library takes; tests/test_gpu_parity.py checks the same identities on the GPU."""
import os
import socket
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, mode, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "ray-tracer_amd"))
    import kkoracle
    from kkdirect import scenes
    o = kkoracle.Oracle(scenes.cornell_box(40, 32))
    if mode == "tile":
        film, st = o.render(4, seed=9, threads=1, tile_index=rank, tile_count=world, block_size=8)
    else:
        film, st = o.render(2, seed=9, threads=1, sample_first=2 * rank)
    t = torch.from_numpy(film)
    dist.reduce(t, dst=0)                       # the film merge (Film::put of every block)
    n = torch.tensor([st.samples], dtype=torch.int64)
    dist.all_reduce(n)
    if rank == 0:
        np.save(out_path, t.numpy())
        assert int(n.item()) == 40 * 32 * 4
    dist.barrier()
    dist.destroy_process_group()


def _run(mode, tmp_path):
    out = str(tmp_path / ("film_%s.npy" % mode))
    mp.spawn(_worker, args=(2, _free_port(), mode, out), nprocs=2, join=True)
    import sys
    import kkoracle
    from kkdirect import scenes
    full, _ = kkoracle.Oracle(scenes.cornell_box(40, 32)).render(4, seed=9, threads=1)
    np.testing.assert_allclose(np.load(out), full, rtol=1e-5, atol=1e-6)


def test_tile_sharded_two_ranks(tmp_path):
    _run("tile", tmp_path)


def test_sample_split_two_ranks(tmp_path):
    _run("sample", tmp_path)
