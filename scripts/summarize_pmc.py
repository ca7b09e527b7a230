"""Summarise the PMC passes of scripts/profile_pmc.sh (gpurun_out/pmc_<workload>/) into profiles/:
   profiles/pmc_traffic.json           per workload: rays per launch + per kernel stage HBM bytes per launch, L2 hit rate, issue counters
   profiles/<tag>_pmc_<workload>.txt   the raw per-stage counter sums

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB and on gfx950 FETCH_SIZE tallies
a 128-byte request as 64 bytes (MI355X_MICROARCH.md, HBM section); FETCH_SIZE includes Infinity-Cache hits.
A stage's "launch" is a kernel launch of that stage (the two-round traversals are two launches per batch), the same
unit bench.py's stage_launches counts.

usage: python scripts/summarize_pmc.py <workload> <tag> [<pmc dir>]"""
import csv, glob, json, os, sys
from collections import defaultdict
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wl = sys.argv[1]; tag = sys.argv[2]
src = sys.argv[3] if len(sys.argv) > 3 else os.path.join(R, "gpurun_out", "pmc_" + wl)
dst = os.path.join(R, "profiles"); os.makedirs(dst, exist_ok=True)
STAGES = ("primary", "shade", "bsdf", "shadow", "resolve")


def stage_of(name):
    if "kPrimary" in name: return "primary"
    if "kShade" in name: return "shade"
    if "kResolve" in name: return "resolve"
    if "kTrace<false" in name or "kTraceRound<false" in name or "kTraceTail<false" in name: return "bsdf"
    if "kTrace<true" in name or "kTraceRound<true" in name or "kTraceTail<true" in name: return "shadow"
    return None


def pmc(dirname):
    """-> {stage: {counter: [value per dispatch]}}; one value per (dispatch, counter), dimensions summed"""
    per = defaultdict(lambda: defaultdict(float)); stage_of_dispatch = {}
    f = max(glob.glob(os.path.join(src, dirname, "*", "*counter_collection.csv")), key=os.path.getmtime)   # the latest run (older ones may linger)
    for row in csv.DictReader(open(f)):
        st = stage_of(row["Kernel_Name"])
        if st is None: continue
        d = row["Dispatch_Id"]; stage_of_dispatch[d] = st
        per[d][row["Counter_Name"]] += float(row["Counter_Value"])
    acc = defaultdict(lambda: defaultdict(list))
    for d, ctrs in per.items():
        for c, v in ctrs.items(): acc[stage_of_dispatch[d]][c].append(v)
    return acc


fetch, write, l2, sq, ta = pmc("fetch"), pmc("write"), pmc("l2"), pmc("sq"), pmc("ta")
bench = json.load(open(os.path.join(src, "fetch.json")))
stages = {}; raw = []
for st in STAGES:
    fs, ws = fetch[st]["FETCH_SIZE"], write[st]["WRITE_SIZE"]
    n = max(1, len(fs))
    f_avg, w_avg = sum(fs) / n, sum(ws) / max(1, len(ws))
    hit, miss = sum(l2[st]["TCC_HIT_sum"]), sum(l2[st]["TCC_MISS_sum"])
    s = {k: sum(v) for k, v in sq[st].items()}; t = {k: sum(v) for k, v in ta[st].items()}
    rec = {"launches_sampled": len(fs), "FETCH_SIZE_KiB_per_launch": round(f_avg, 1), "WRITE_SIZE_KiB_per_launch": round(w_avg, 1),
           "hbm_bytes_per_launch": round((2 * f_avg + w_avg) * 1024), "l2_hit_rate": round(hit / max(1.0, hit + miss), 4),
           "ea_rdreq_32B_share": round(sum(l2[st]["TCC_EA0_RDREQ_32B_sum"]) / max(1.0, sum(l2[st]["TCC_EA0_RDREQ_sum"])), 4)}
    if s.get("SQ_WAVES"):
        rec.update({"valu_insts_per_wave": round(s["SQ_INSTS_VALU"] / s["SQ_WAVES"], 1), "salu_insts_per_wave": round(s["SQ_INSTS_SALU"] / s["SQ_WAVES"], 1),
                    "vmem_rd_insts_per_wave": round(s["SQ_INSTS_VMEM_RD"] / s["SQ_WAVES"], 1),
                    # share of the cycles a wave is resident in which it waits on anything
                    "wave_wait_share": round(s["SQ_WAIT_INST_ANY"] / max(1.0, s["SQ_WAVE_CYCLES"]), 4),
                    # wave-level VALU instructions per kernel launch of the stage: bench.py divides by its own launch duration and by the
                    # chip's issue peak (1024 SIMDs x one wave64 VALU instruction per 4 cycles)
                    "valu_insts_per_launch": round(s["SQ_INSTS_VALU"] / max(1, len(sq[st]["SQ_INSTS_VALU"])))})
    if t.get("GRBM_GUI_ACTIVE"):
        rec["ta_busy_share"] = round(t.get("TA_BUSY_avr", 0.0) / max(1.0, t["GRBM_GUI_ACTIVE"]), 4)
    stages[st] = rec
    raw.append("%s %s" % (st, {**{k: "%.4g" % v for k, v in sorted(s.items())}, **{k: "%.4g" % v for k, v in sorted(t.items())},
                               **{k: "%.4g" % sum(v) for k, v in sorted(l2[st].items())}}))
# the key bench.py matches on: camera samples per batch of the profiled render (the bench line of the fetch pass)
path = os.path.join(dst, "pmc_traffic.json")
allrec = json.load(open(path)) if os.path.exists(path) else {}
allrec = {k: v for k, v in allrec.items() if isinstance(v, dict) and "stages" in v}      # drop the round-1 layout
allrec[wl] = {"samples_per_batch": bench["pmc_samples_per_batch"], "spp_profiled": bench["config"]["spp_total"], "workload": bench["config"]["workload"],
              "note": "one lane (KKD_STREAMS=1), separate rocprofv3 --pmc passes (scripts/profile_pmc.sh), (2*FETCH_SIZE + WRITE_SIZE)*1024 per kernel launch",
              "stages": stages}
json.dump(allrec, open(path, "w"), indent=1)
open(os.path.join(dst, "%s_pmc_%s.txt" % (tag, wl)), "w").write("\n".join(raw) + "\n")
print(json.dumps(allrec[wl], indent=1))
