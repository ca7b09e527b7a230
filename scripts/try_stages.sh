#!/bin/bash
# On the GPU box: one-lane stage times per C4 step of several builds (arguments: library paths).
set -u
mkdir -p gpurun_out
for lib in "$@"; do
  export KKD_LIB=$lib
  timeout -k 10 400 python3 bench.py --workload c4 --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/stages.json 2> gpurun_out/stages.err || { echo "bench failed"; tail -5 gpurun_out/stages.err; exit 1; }
  python3 -c "import json; d=json.loads(open('gpurun_out/stages.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$lib', d['value'], r['one_lane_ms_per_step'], r['stage_ms_per_step'])"
done
