#!/bin/bash
# On the GPU box: C4 / C2 throughput against the number of camera samples in flight per batch.
set -u
mkdir -p gpurun_out
for wl in c4 c2; do for b in 66355200 132710400 265420800; do
  timeout -k 10 300 python3 bench.py --workload $wl --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-extra-passes --no-secondary > gpurun_out/batch_${wl}_$b.json 2> gpurun_out/batch_${wl}_$b.err || { echo "bench $wl $b failed"; tail -5 gpurun_out/batch_${wl}_$b.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/batch_${wl}_$b.json').read().strip().splitlines()[-1]); print('$wl batch $b', d['value'], d['ms_per_step'])"
done; done
