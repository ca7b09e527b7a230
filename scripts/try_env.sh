#!/bin/bash
# On the GPU box: throughput of the workloads in $WLS (default "c4 c2") under several environment settings (arguments: "VAR=val VAR2=val" strings), no tests.
set -u
mkdir -p gpurun_out
for setting in "$@"; do
  for wl in ${WLS:-c4 c2}; do
    env KKD_DEBUG_ROUNDS=1 $setting timeout -k 10 300 python3 bench.py --workload $wl --steps ${STEPS:-6} --warmup ${WARMUP:-2} --no-cpu-baseline --no-extra-passes --no-secondary > gpurun_out/env_${wl}.json 2> gpurun_out/env_${wl}.err || { echo "bench $wl failed"; tail -5 gpurun_out/env_${wl}.err; exit 1; }
    python3 -c "import json,sys; d=json.loads(open('gpurun_out/env_${wl}.json').read().strip().splitlines()[-1]); print('$setting $wl', d['value'], d['ms_per_step'])"
    grep "^rounds:" gpurun_out/env_${wl}.err | tail -2 | tr '\n' ' '; echo
  done
done
