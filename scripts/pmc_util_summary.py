import csv, glob, sys, os
from collections import defaultdict
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def stage_of(name):
    if "<true" in name and "kPrimary" in name: return None
    if "kTrace<false, true" in name or "kTrace<true, true" in name or "kTraceRound<false, true" in name or "kTraceRound<true, true" in name or "kTraceTail<false, true" in name or "kTraceTail<true, true" in name: return None
    for k, v in (("kPrimary", "primary"), ("kShade", "shade"), ("kResolve", "resolve"), ("kTrace<false", "bsdf"), ("kTrace<true", "shadow"), ("kTraceRound<false", "bsdf"), ("kTraceRound<true", "shadow"), ("kTraceTail<false", "bsdf"), ("kTraceTail<true", "shadow")):
        if k in name: return v
acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for d in ("sq", "ta"):
    for f in glob.glob(os.path.join(R, "gpurun_out", "pmc_util", d, "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            st = stage_of(row["Kernel_Name"])
            if st: acc[st][row["Counter_Name"]].append(float(row["Counter_Value"]))
for st in ("primary", "shade", "bsdf", "shadow", "resolve"):
    print(st, {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(acc[st].items())})
