#!/bin/bash
# Run on the GPU box (via gpurun): one PMC pass for the active-lane share of VALU instructions per kernel.
#   scripts/profile_lanes.sh <workload> <spp> <outdir>
set -u
WL=${1:-c4}; SPP=${2:-32}; OUT=${3:-gpurun_out/lanes_$WL}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $OUT; mkdir -p $OUT
export KKD_STREAMS=1
ARGS="bench.py --workload $WL --steps 1 --warmup 0 --spp $SPP --no-cpu-baseline --no-extra-passes"
timeout -s KILL 400 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/lanes -- python3 $ARGS > $OUT/lanes.json 2> $OUT/lanes.err || { echo "lanes failed"; tail -5 $OUT/lanes.err; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/lanes/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:12]:
    a = v.get("SQ_ACTIVE_INST_VALU", 0); t = v.get("SQ_THREAD_CYCLES_VALU", 0); i = v.get("SQ_INSTS_VALU", 0)
    print(f"{k:60s} insts {i:.3e} active {a:.3e} thread_cycles {t:.3e} lanes/inst {t / max(a, 1):.1f} ({t / max(i, 1):.1f})")
PY
