#!/bin/bash
# On the GPU box: C4 / C2 throughput under several environment settings (arguments: "VAR=val VAR2=val" strings), no tests.
set -u
mkdir -p gpurun_out
for setting in "$@"; do
  for wl in c4 c2; do
    env $setting timeout -k 10 300 python3 bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-extra-passes --no-secondary > gpurun_out/env_${wl}.json 2> gpurun_out/env_${wl}.err || { echo "bench $wl failed"; tail -5 gpurun_out/env_${wl}.err; exit 1; }
    python3 -c "import json,sys; d=json.loads(open('gpurun_out/env_${wl}.json').read().strip().splitlines()[-1]); print('$setting $wl', d['value'], d['ms_per_step'])"
  done
done
