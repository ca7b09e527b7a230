"""Sweep of the traversal-round knobs on one workload (run on the GPU box): the scene is generated once and handed to one
child process per setting (the knobs are read from the environment once per process).
   python scripts/sweep_rounds2.py c4 64 "KKD_TRACE_ROUNDS=3 KKD_ITER_CAP_LATER=48" "..." """
import os, pickle, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "ray-tracer_amd")); sys.path.insert(0, R)
CHILD = r'''
import os, pickle, sys, time
import torch      # before libkkdirect: the HIP runtime loaded first serves both
sys.path.insert(0, os.path.join(%r, "ray-tracer_amd"))
import kkdirect
sc = pickle.load(open("/tmp/sweep_scene.pkl", "rb"))
ctx = kkdirect.Context(device=0, seed=1, one_lane=bool(int(os.environ.get("SWEEP_ONE_LANE", "1")))).upload(sc)
film = torch.zeros(ctx.height * ctx.width * 5, dtype=torch.float32, device="cuda")
spp = int(sys.argv[1])
for _ in range(3): ctx.render_device(film.data_ptr(), spp)
st = [ctx.render_device(film.data_ptr(), spp) for _ in range(3)]
ms = lambda k: sum(s.stage_seconds[k] for s in st) / 3 * 1e3
print("%%-60s total %%8.2f ms  primary %%7.2f shade %%7.2f bsdf %%7.2f shadow %%7.2f resolve %%7.2f" %% (os.environ.get("SWEEP_LABEL", "-"), sum(s.seconds for s in st) / 3 * 1e3, ms(0), ms(1), ms(2), ms(3), ms(4)), flush=True)
''' % R
if __name__ == "__main__":
    import bench
    wl, spp = sys.argv[1], sys.argv[2]
    scene, _, label = bench.make_workload(wl, 1.0)
    scene._keep = None
    pickle.dump(scene, open("/tmp/sweep_scene.pkl", "wb"), protocol=4)
    print(label, "spp", spp, flush=True)
    for setting in sys.argv[3:]:
        env = dict(os.environ, SWEEP_LABEL=setting)
        for kv in setting.split():
            if "=" in kv:
                k, v = kv.split("=", 1); env[k] = v
        p = subprocess.run([sys.executable, "-c", CHILD, spp], env=env, capture_output=True, text=True, timeout=300)
        if os.environ.get("SWEEP_STDERR"):
            sys.stdout.write("\n".join(l for l in p.stderr.splitlines() if l.startswith("rounds:"))[-400:] + "\n")
        sys.stdout.write(p.stdout if p.returncode == 0 else "FAILED %s: %s\n" % (setting, p.stderr[-500:])); sys.stdout.flush()
        if p.returncode != 0:
            sys.exit(1)
