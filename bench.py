#!/usr/bin/env python3
"""bench.py -- throughput of the kk direct-lighting path on MI355X (BASELINE.json metric: Msamples/s and
Mrays/s, fraction of the HBM roofline, CPU baseline beside it).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

One step = one full render pass of the workload.  Default = C4 of BASELINE.json, the largest single-GPU
configuration and the north-star target: 10.7 M-triangle displaced mesh under a 4096x2048 HDR envmap,
roughconductor / dielectric, 3840x2160, 512 spp, bsdfSamples = lightSamples = 1 (C2, the Sponza-class
1080p scene, is timed briefly in the same run and reported under "secondary").  With N ranks the image
tiles are sharded round-robin over the ranks (tile_index = rank, tile_count = N, no collective while
rendering) and every rank renders N x spp of its tiles, so per-GPU work is constant (weak scaling); the
films are merged by one RCCL sum-reduce per step (what Film::put(block) does in the reference), inside the
timed region.  --split sample renders the whole frame on every rank with disjoint sample ranges instead.

Roofline (dominant kernel = the kernel stage with the largest exclusive time): ONE object, every use figure <= 1 by
construction, all of it reproducible from profiles/:
  valu_issue.frac = time the chip's 1024 SIMDs need to ISSUE the kernel's VALU instructions / the launch duration measured in
      this run -- instructions per launch from PMC SQ_INSTS_VALU (profiles/pmc_traffic.json), weighted class by class
      (profiles/r3_isa_classes.json: instruction mix of the traversal loop) with the issue rates tests/tools/valu_peak.hip
      measured on this chip (profiles/valu_peak_gfx950.txt);
  hbm.frac        = HBM bytes per launch from the PMC passes ((2 x FETCH_SIZE + WRITE_SIZE) x 1024, gfx950 correction of
      MI355X_MICROARCH.md) / duration / 8 TB/s;
  bound / achieved / peak / frac = the larger of the two, named.
  algorithmic_gbps = SURVEY 8d's accounting figure (rays x (io + nodes/ray x node bytes + tris/ray x triangle bytes), the
      per-ray counts measured by a counting pass of the same traversal in this run) / duration.  It prices every node and
      triangle visit as if it came from HBM; the caches serve most of them, so it exceeds the HBM peak by design and has NO
      peak attached.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
# unit of the issue rates in profiles/valu_peak_gfx950.txt: wave64 VALU instructions per SIMD per 4 cycles at 2.4 GHz, 256 CUs x 4 SIMDs
VALU_SLOTS_PER_S = 256 * 4 * 2.4e9 / 4


def load_valu_model(stage):
    """-> (seconds the chip needs to issue ONE wave-level VALU instruction of this stage's mix, detail) from the committed
    microbenchmark rates and the loop's instruction mix; None when either file is missing"""
    import re
    try:
        rates = {}
        for line in open(os.path.join(ROOT, "profiles", "valu_peak_gfx950.txt")):
            m = re.match(r"(.{18})\s+[\d.]+ ms\s+[\d.e+]+ wave-instructions/s\s+=\s+([\d.]+) per SIMD", line)
            if m:
                rates[m.group(1).strip()] = float(m.group(2))
        mix = json.load(open(os.path.join(ROOT, "profiles", "r3_isa_classes.json")))[stage]["share_by_kind"]
        slots = sum(share / rates[kind] for kind, share in mix.items())       # SIMD issue slots (4 cycles) per instruction of the mix
        return slots / VALU_SLOTS_PER_S, {"issue_slots_per_instruction": round(slots, 4), "mix": mix,
                                          "rates_per_simd_per_4_cycles": {k: rates[k] for k in mix}}
    except Exception:
        return None


def make_workload(name, scale):
    from kkdirect import scenes
    if name == "c1":
        return scenes.cornell_box(512, 512), 64, "C1 Cornell box 512x512 64spp"
    if name == "c2":
        w, h = int(1920 * scale), int(1080 * scale)
        return scenes.sponza_class(w, h), 256, "C2 Sponza-class interior ~260k tris %dx%d 256spp" % (w, h)
    if name == "c3":
        w, h = int(1920 * scale), int(1080 * scale)
        return scenes.many_lights(w, h, interior_tris=260_000), 128, "C3 many-light: C2 geometry + 10k emissive triangles %dx%d 128spp" % (w, h)
    if name == "c4":
        w, h = int(3840 * scale), int(2160 * scale)
        return scenes.scanned_mesh(w, h), 512, "C4 10.5M-tri mesh + HDR envmap %dx%d 512spp" % (w, h)
    raise SystemExit("unknown workload " + name)


def stage_bytes(st, n_b, n_l, per_ray, node_bytes=64.0, tri_bytes=48.0):
    """Algorithmic bytes per kernel stage for one render ('algorithmic bytes').
    per_ray[k] = (nodes/ray, tris/ray) measured by the counting build of the same traversal."""
    rays = list(st.class_rays)
    io = (16.0, 48.0, 36.0)   # primary: hit write; bsdf: ray read + hit write; shadow: ray read + flag write
    trav = [rays[k] * (io[k] + per_ray[k][0] * node_bytes + per_ray[k][1] * tri_bytes) for k in range(3)]
    s = float(st.samples)
    # round 3 record layout: one origin per sample, direction planes, flag bytes; accum only where it is not the default
    shade = s * (16 + tri_bytes + 12 + 36 + 32 + 128 + 64 + 12 + 36 + 16 + 1 + n_b * (16 + 16 + 1) + n_l * (16 + 16))
    resolve = s * (1 + 16 + n_b * (16 + 16 + 16 + 1) + n_l * 20 + 125.0 / 64.0 * 8.0)
    return [trav[0], shade, trav[1], trav[2], resolve]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c4", choices=["c1", "c2", "c3", "c4"])
    ap.add_argument("--no-extra-passes", action="store_true", help="timed region only: no counting pass, no one-lane pass, no secondary workload (profiler runs)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short C2 measurement reported under 'secondary'")
    ap.add_argument("--split", default="tile", choices=["tile", "sample"])
    ap.add_argument("--spp", type=int, default=0, help="override the workload's spp (then the line is not the headline config)")
    ap.add_argument("--scale", type=float, default=1.0, help="resolution scale (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--leaf-test", default="projection", choices=["watertight", "projection"],
                    help="projection (the library's default) = the reference's TriAccel arithmetic, bit-identical hit ids; watertight = KKD_FLAG_WATERTIGHT_TEST")
    ap.add_argument("--integrator", default="kk", choices=["kk", "direct", "path"],
                    help="kk = the headline path; direct / path = the sibling plugins on the same kernels (not the headline metric)")
    ap.add_argument("--max-depth", type=int, default=-1, help="path: maxDepth")
    ap.add_argument("--rr-depth", type=int, default=5, help="path: rrDepth")
    ap.add_argument("--batch", type=int, default=0, help="camera samples in flight per batch (0 = library default)")
    args = ap.parse_args()

    # the host driver of this pool only supports dmabuf IPC: without this RCCL's peer setup fails with "hipIpcGetMemHandle: invalid
    # argument" (the launcher's environment normally carries it already)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    import torch
    import kkdirect
    from kkdirect import abi
    if not torch.cuda.is_available() or kkdirect.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: libkkdirect has no CPU fallback")
    # $KKD_BENCH_BACKEND=gloo rehearses the N-rank path on a box with fewer GPUs than ranks (ranks share devices, the film
    # merge is an all_reduce because gloo has no CUDA reduce); the measured configuration is always nccl = RCCL
    backend = os.environ.get("KKD_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    scene, spp, label = make_workload(args.workload, args.scale)
    if args.spp:
        spp = args.spp
    n_b = n_l = 1
    t0 = time.perf_counter()
    proj = args.leaf_test == "projection"
    # One process, several GPUs (python bench.py --gpus N without a launcher): the library's own multi-device path
    # (kkd_multi_*: one host thread per device, tile gather over xGMI or RCCL reduce).  Under torch.distributed.run every
    # rank is one process with one GPU and the films are reduced through torch (RCCL) below.
    in_library = world == 1 and args.gpus > 1
    if in_library:
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit("--gpus %d but %d devices visible" % (args.gpus, torch.cuda.device_count()))
        ctx = kkdirect.MultiContext(list(range(args.gpus)), split=args.split, seed=1, max_samples_in_flight=args.batch, projection_test=proj).upload(scene)
    else:
        ctx = kkdirect.Context(device=local_rank, seed=1, max_samples_in_flight=args.batch, projection_test=proj).upload(scene)
    upload_s = time.perf_counter() - t0
    W, H = ctx.width, ctx.height
    film = torch.zeros(H * W * 5, dtype=torch.float32, device="cuda")

    if in_library:
        kw = dict(spp=spp * args.gpus)          # the library shards it; per-GPU work stays constant (weak scaling)
    elif args.split == "tile":
        kw = dict(spp=spp * world, tile_index=rank, tile_count=world, sample_first=0)
    else:
        kw = dict(spp=spp, tile_index=0, tile_count=1, sample_first=rank * spp)
    if args.integrator != "kk":
        kw.update(integrator={"direct": abi.INTEGRATOR_DIRECT, "path": abi.INTEGRATOR_PATH}[args.integrator])
        if args.integrator == "path":
            kw.update(max_depth=args.max_depth, rr_depth=args.rr_depth)

    def step():
        st = ctx.render_device(film.data_ptr(), bsdf_samples=n_b, light_samples=n_l, **kw)
        if dist is not None:
            if backend == "nccl":
                dist.reduce(film, dst=0)      # film merge = Film::put(block): sum of per-rank films
            else:
                dist.all_reduce(film)
        return st

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    fence()
    elapsed = time.perf_counter() - t0
    samples = sum(s.samples for s in stats)
    rays = sum(s.rays_closest + s.rays_shadow for s in stats)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([samples, rays], dtype=torch.float64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        samples, rays = float(c[0].item()), float(c[1].item())

    film_weight = float(film.view(-1, 5)[:, 4].sum().item()) if rank == 0 else 0.0
    out = None
    if rank == 0:
        roofline = None
        if not args.no_extra_passes and not in_library:
            roofline = measure_roofline(ctx, scene, args, kw, n_b, n_l, spp, film, stats, elapsed, local_rank, proj, W, H)
        cpu = None
        if not args.no_cpu_baseline and world == 1 and not in_library and not args.no_extra_passes:
            cpu = cpu_baseline(scene, n_b, n_l, W, H)
        secondary = None
        if world == 1 and not in_library and args.workload == "c4" and args.integrator == "kk" and not (args.no_secondary or args.no_extra_passes or args.spp or args.scale != 1.0):
            secondary = secondary_c2(local_rank, proj)
        out = {
            "metric": "Msamples/s (kk direct integrator)" if args.integrator == "kk" else "Msamples/s (%s plugin, not the headline metric)" % args.integrator, "value": round(samples / elapsed / 1e6, 3), "unit": "Msamples/s",
            "n_gpus": args.gpus if in_library else world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "mrays_per_s": round(rays / elapsed / 1e6, 3),
            "config": {"workload": label, "spp_total": kw["spp"] if (args.split == "tile" or in_library) else spp * world,
                       "multi_gpu": "in-library (kkd_multi, merge: %s)" % ctx.merge_path if in_library else ("one process per GPU, film reduce through torch.distributed (%s)" % backend if world > 1 else None),
                       # weight channel of the merged film / samples of one step: 1 up to what the filter spills over the frame edge
                       "film_weight_per_sample": round(film_weight / max(1.0, samples / args.steps), 5), "bsdfSamples": n_b, "lightSamples": n_l,
                       "integrator": args.integrator, "max_depth": args.max_depth if args.integrator == "path" else None,
                       "split": args.split, "backend": backend, "leaf_test": args.leaf_test, "triangles": int(scene._nt), "scene_upload_s": round(upload_s, 2)},
            "roofline": roofline, "cpu_baseline": cpu, "secondary": secondary,
        }
        if args.no_extra_passes:     # profiler runs: what scripts/summarize_pmc.py keys the counters by
            out["pmc_samples_per_batch"] = round(samples / max(1, sum(s.stage_launches[1] for s in stats)))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


def load_pmc(workload, stage, samples_per_batch):
    """HBM traffic / L2 hit rate / issue counters of one kernel stage from the committed PMC summary
    (profiles/pmc_traffic.json, written by scripts/summarize_pmc.py from separate rocprofv3 --pmc passes).  Only
    returned when the passes were taken on this workload with the same number of camera samples per batch (within 2 %):
    traffic per launch scales with the batch, not with spp."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path)).get(workload)
        if not rec or abs(rec["samples_per_batch"] - samples_per_batch) > 0.02 * samples_per_batch:
            return None
        return rec["stages"].get(stage)
    except Exception:
        return None


def measure_roofline(ctx, scene, args, kw, n_b, n_l, spp, film, stats, elapsed, device, proj, W, H):
    """Roofline of the dominant kernel stage.  Durations: HIP events the library records between the kernels, on the
    stream they run on.  In the timed region two batches of camera samples are in flight on two streams, so an event
    interval there includes the other lane's kernels; the kernel's exclusive duration comes from a short one-lane pass
    (KKD_FLAG_ONE_LANE: same kernels, same batches, one stream) of the same render right here."""
    import kkdirect
    from kkdirect import abi
    # nodes / triangles per ray of each ray class: counting build of the same traversal, low spp
    cctx = kkdirect.Context(device=device, seed=1, count_traversal=True, projection_test=proj).upload(scene)
    _, cst = cctx.render(min(spp, 8), n_b, n_l, **{k: v for k, v in kw.items() if k not in ("spp", "sample_first")})
    cctx.close()
    per_ray = [(cst.class_nodes[k] / max(1, cst.class_rays[k]), cst.class_tris[k] / max(1, cst.class_rays[k])) for k in range(3)]
    shared_s = [sum(s.stage_seconds[k] for s in stats) for k in range(5)]
    shared_n = [sum(s.stage_launches[k] for s in stats) for k in range(5)]
    x_steps = max(1, min(args.steps, 3))
    xctx = kkdirect.Context(device=device, seed=1, max_samples_in_flight=args.batch, projection_test=proj, one_lane=True).upload(scene)
    xctx.render_device(film.data_ptr(), bsdf_samples=n_b, light_samples=n_l, **kw)          # warm-up (budgets adapt)
    xstats = [xctx.render_device(film.data_ptr(), bsdf_samples=n_b, light_samples=n_l, **kw) for _ in range(x_steps)]
    one_lane_ms = sum(s.seconds for s in xstats) / x_steps * 1e3
    xctx.close()
    stage_s = [sum(s.stage_seconds[k] for s in xstats) for k in range(5)]
    stage_n = [sum(s.stage_launches[k] for s in xstats) for k in range(5)]
    acc = ctx.accel_info()
    bytes_k = [sum(stage_bytes(s, n_b, n_l, per_ray, acc["node_bytes"], acc["tri_bytes"])[k] for s in xstats) for k in range(5)]
    dom = int(np.argmax(stage_s)) if sum(stage_s) > 0 else 2    # the path plugin times the render only, not its stages
    name = abi.STAGE_NAMES[dom]
    launches = max(1, stage_n[dom])
    avg_launch_s = max(stage_s[dom] / launches, 1e-12)
    alg_per_launch = bytes_k[dom] / launches
    ray_class = {0: 0, 2: 1, 3: 2}.get(dom)
    rays_per_launch = None if ray_class is None else sum(s.class_rays[ray_class] for s in xstats) / launches
    samples_per_batch = sum(s.samples for s in xstats) / max(1, stage_n[1])      # kShade runs once per batch
    pmc = None if args.scale != 1.0 else load_pmc(args.workload, name, samples_per_batch)    # another resolution is another workload
    traffic = None if pmc is None else pmc["hbm_bytes_per_launch"]
    per_batch = launches / max(1, stage_n[1])
    kernel_note = {"bsdf": "bsdf = kTraceRound<closest> + kTraceTail<closest> (%d rounds = %d launches per batch of rays, averaged)" % (per_batch, per_batch),
                   "shadow": "shadow = kTraceRound<any> + kTraceTail<any> (%d rounds = %d launches per batch of rays, averaged)" % (per_batch, per_batch)}
    alg_gbps = alg_per_launch / avg_launch_s / 1e9
    hbm = None
    if traffic is not None:
        g = traffic / avg_launch_s / 1e9
        hbm = {"achieved": round(g, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(g / HBM_PEAK_GBS, 5), "bytes_per_launch": traffic,
               "l2_hit_rate": pmc.get("l2_hit_rate"),
               "source": "PMC (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch, profiles/pmc_traffic.json; FETCH_SIZE includes Infinity-Cache hits: an upper bound of the DRAM traffic"}
    valu = None
    model = load_valu_model(name) if pmc is not None and pmc.get("valu_insts_per_launch") else None
    if model is not None:
        sec_per_inst, detail = model
        n = pmc["valu_insts_per_launch"]
        t_issue = n * sec_per_inst
        valu = {"achieved": round(n / avg_launch_s / 1e9, 2), "peak": round(1.0 / sec_per_inst / 1e9, 2), "unit": "G wave-instructions/s",
                "frac": round(t_issue / avg_launch_s, 5), "valu_instructions_per_launch": n,
                "issue_time_ms_per_launch": round(t_issue * 1e3, 4), **detail,
                "source": "PMC SQ_INSTS_VALU per launch (profiles/pmc_traffic.json) x the loop's instruction mix (profiles/r3_isa_classes.json) / the per-class issue rates measured by tests/tools/valu_peak.hip (profiles/valu_peak_gfx950.txt)"}
    # the bound is the resource in larger use; both fractions are <= 1 by construction
    if valu is not None and (hbm is None or valu["frac"] >= hbm["frac"]):
        top = {"bound": "valu-issue", "achieved": valu["achieved"], "peak": valu["peak"], "unit": valu["unit"], "frac": valu["frac"]}
    elif hbm is not None:
        top = {"bound": "hbm", "achieved": hbm["achieved"], "peak": hbm["peak"], "unit": hbm["unit"], "frac": hbm["frac"]}
    else:
        top = {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None}
    r = {**top, "kernel": kernel_note.get(name, name), "traffic": traffic, "valu_issue": valu, "hbm": hbm,
         "pmc_source": None if pmc is None else "profiles/pmc_traffic.json (separate rocprofv3 --pmc passes, same workload and batch size)",
         # SURVEY 8d's accounting figure: no peak attached (cache-served visits priced as HBM bytes; exceeds the HBM peak by design)
         "algorithmic_gbps": round(alg_gbps, 2), "algorithmic_bytes_per_launch": round(alg_per_launch),
         "avg_launch_ms": round(avg_launch_s * 1e3, 4), "launches": stage_n[dom], "rays_per_launch": None if rays_per_launch is None else round(rays_per_launch), "samples_per_batch": round(samples_per_batch),
         "measured": "one-lane pass (KKD_FLAG_ONE_LANE) of the same render inside this run, %d steps: exclusive kernel durations" % x_steps,
         "avg_launch_ms_timed_region": round(shared_s[dom] / max(1, shared_n[dom]) * 1e3, 4),
         "one_lane_ms_per_step": round(one_lane_ms, 3),
         "nodes_per_ray": [round(p[0], 2) for p in per_ray], "tris_per_ray": [round(p[1], 2) for p in per_ray],
         "node_bytes": acc["node_bytes"], "tri_bytes": acc["tri_bytes"],
         # whole job: algorithmic bytes of all five stages per second of the timed region, and the
         # compulsory lower bound (scene structures read once + film written once) the re-reads are measured against
         "job_algorithmic_gbps": round(sum(sum(stage_bytes(s_, n_b, n_l, per_ray, acc["node_bytes"], acc["tri_bytes"])) for s_ in stats) / elapsed / 1e9, 2),
         "compulsory_bytes_per_step": int(acc["n_nodes"] * (acc["node_bytes"] + 128) + acc["n_leaf_tris"] * acc["tri_bytes"]
                                          + scene._nv * (12 + 12 + 8) + scene._nt * 12 + W * H * 5 * 4),
         "stage_ms_per_step": {abi.STAGE_NAMES[k]: round(stage_s[k] / x_steps * 1e3, 3) for k in range(5)},
         "stage_ms_per_step_timed_region": {abi.STAGE_NAMES[k]: round(shared_s[k] / args.steps * 1e3, 3) for k in range(5)}}
    return r


def secondary_c2(device, proj):
    """C2 (Sponza-class interior, 1080p, 256 spp) timed briefly in the same run: the cache-resident counterpart of the headline."""
    import torch
    import kkdirect
    scene, spp, label = make_workload("c2", 1.0)
    ctx = kkdirect.Context(device=device, seed=1, projection_test=proj).upload(scene)
    film = torch.zeros(ctx.height * ctx.width * 5, dtype=torch.float32, device="cuda")
    for _ in range(2):
        ctx.render_device(film.data_ptr(), spp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = [ctx.render_device(film.data_ptr(), spp) for _ in range(5)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.close()
    return {"workload": label, "value": round(sum(s.samples for s in st) / dt / 1e6, 3), "unit": "Msamples/s", "steps": 5, "warmup": 2,
            "ms_per_step": round(dt / 5 * 1e3, 3), "mrays_per_s": round(sum(s.rays_closest + s.rays_shadow for s in st) / dt / 1e6, 3)}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return None


def cpu_baseline(scene, n_b, n_l, W, H):
    """The oracle (CPU restatement of the reference path: SAH kd-tree + TriAccel + Havran + kk Li, 32x32 tiles,
    one thread per core) timed on a bounded sample of the same workload: the full frame at a low spp."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import kkoracle
    cores = min(16, os.cpu_count() or 1)      # the box's CPU share for one GPU (gpurun sets OMP_NUM_THREADS etc. to 16); os.cpu_count() is the whole machine
    o = kkoracle.Oracle(scene)
    _, s1 = o.render(1, n_b, n_l, threads=cores)
    spp = int(max(1, min(64, 12.0 / max(s1.seconds, 1e-3))))
    _, st = o.render(spp, n_b, n_l, threads=cores)
    try:
        affinity = len(os.sched_getaffinity(0))
    except Exception:
        affinity = None
    return {"value": round(st.samples / st.seconds / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "host_logical_cpus": os.cpu_count(), "host_cpus_in_affinity_mask": affinity, "host_cpu_model": _cpu_model(),
            "mrays_per_s": round((st.rays_closest + st.rays_shadow) / st.seconds / 1e6, 3),
            "sample": "same scene and resolution (%dx%d) at %d spp instead of the workload's spp, %.1f s of CPU time" % (W, H, spp, st.seconds)}


if __name__ == "__main__":
    main()
