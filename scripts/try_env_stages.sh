#!/bin/bash
# On the GPU box: throughput and one-lane stage times per C4 step under several environment settings (arguments: "VAR=val ..." strings).
set -u
mkdir -p gpurun_out
for setting in "$@"; do
  env KKD_DEBUG_ROUNDS=1 $setting timeout -k 10 400 python3 bench.py --workload ${WL:-c4} --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/stages.json 2> gpurun_out/stages.err || { echo "bench failed"; tail -5 gpurun_out/stages.err; exit 1; }
  python3 -c "import json; d=json.loads(open('gpurun_out/stages.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$setting', d['value'], r['one_lane_ms_per_step'], r['stage_ms_per_step'])"
  grep "^rounds:" gpurun_out/stages.err | sed -n '5,6p' | tr '\n' ' '; echo
done
