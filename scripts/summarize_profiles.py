"""Summarise gpurun_out/prof (scripts/profile_r1.sh) into profiles/: per-kernel stats of the bench run and the
per-launch HBM traffic / L2 hit rate per kernel stage from the PMC passes.

Traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes: FETCH_SIZE/WRITE_SIZE are in KiB and on gfx950
FETCH_SIZE reads half of the fetched bytes (MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = os.path.join(R, "gpurun_out", "prof")
dst = os.path.join(R, "profiles"); os.makedirs(dst, exist_ok=True)

def stage_of(name):
    if "kPrimary" in name: return "primary"
    if "kShade" in name: return "shade"
    if "kResolve" in name: return "resolve"
    if "kTrace<false" in name or "kTraceRound<false" in name or "kTraceTail<false" in name: return "bsdf"
    if "kTrace<true" in name or "kTraceRound<true" in name or "kTraceTail<true" in name: return "shadow"
    return None

def counting(name):   # counting variants (COUNT template argument true) are not part of the timed loop
    return "kPrimary<true" in name or "kTrace<false, true" in name or "kTrace<true, true" in name or "kTraceRound<false, true" in name or "kTraceRound<true, true" in name or "kTraceTail<false, true" in name or "kTraceTail<true, true" in name

shutil.copy(glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))[0], os.path.join(dst, "%s_bench_c2_kernel_stats.csv" % tag))
shutil.copy(os.path.join(src, "bench_under_trace.json"), os.path.join(dst, "%s_bench_c2_under_rocprof.json" % tag))
shutil.copy(glob.glob(os.path.join(src, "trace1", "*", "*kernel_stats.csv"))[0], os.path.join(dst, "%s_bench_c2_one_lane_kernel_stats.csv" % tag))
shutil.copy(os.path.join(src, "bench_under_trace1.json"), os.path.join(dst, "%s_bench_c2_one_lane_under_rocprof.json" % tag))

def pmc(dirname):
    acc = defaultdict(lambda: defaultdict(list))
    f = glob.glob(os.path.join(src, dirname, "*", "*counter_collection.csv"))[0]
    for row in csv.DictReader(open(f)):
        st = stage_of(row["Kernel_Name"])
        if st is None or counting(row["Kernel_Name"]): continue
        acc[st][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc

fetch, write, l2 = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_l2")
out = {}
for st in ("primary", "shade", "bsdf", "shadow", "resolve"):
    fs = fetch[st]["FETCH_SIZE"]; ws = write[st]["WRITE_SIZE"]
    hit = sum(l2[st]["TCC_HIT_sum"]); miss = sum(l2[st]["TCC_MISS_sum"])
    f_avg = sum(fs) / max(1, len(fs)); w_avg = sum(ws) / max(1, len(ws))
    out[st] = {"launches_sampled": len(fs), "FETCH_SIZE_KiB_per_launch": round(f_avg, 1), "WRITE_SIZE_KiB_per_launch": round(w_avg, 1),
               "hbm_bytes_per_launch": round((2 * f_avg + w_avg) * 1024), "l2_hit_rate": round(hit / max(1.0, hit + miss), 4)}
json.dump({"c2": {k: v["hbm_bytes_per_launch"] for k, v in out.items()}, "_detail_c2": out,
           "_note": "per launch of one 16.6M-sample batch (bench.py --spp 32 gives the batch size of the headline run), one lane; (2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes, scripts/profile_r1.sh"},
          open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
print(open(os.path.join(dst, "%s_bench_c2_kernel_stats.csv" % tag)).read()[:1500])
