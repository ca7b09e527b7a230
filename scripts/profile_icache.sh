#!/bin/bash
# Run on the GPU box (via gpurun): one PMC pass for the instruction-cache behaviour per kernel.   scripts/profile_icache.sh <workload> <spp> <outdir>
set -u
WL=${1:-c4}; SPP=${2:-32}; OUT=${3:-gpurun_out/icache_$WL}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $OUT; mkdir -p $OUT
export KKD_STREAMS=1
ARGS="bench.py --workload $WL --steps 1 --warmup 0 --spp $SPP --no-cpu-baseline --no-extra-passes"
timeout -s KILL 400 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/ic -- python3 $ARGS > $OUT/ic.json 2> $OUT/ic.err || { echo "icache pass failed"; tail -5 $OUT/ic.err; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/ic/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:9]:
    print(f"{k:60s} " + " ".join(f"{c}={v[c]:.3e}" for c in sorted(v)))
PY
