#!/bin/bash
# On the GPU box: C4 / C2 throughput with the reference's projection leaf test against the watertight default.
set -u
mkdir -p gpurun_out
for wl in c4 c2; do for lt in watertight projection; do
  timeout -k 10 300 python3 bench.py --workload $wl --leaf-test $lt --steps 6 --warmup 2 --no-cpu-baseline --no-extra-passes --no-secondary > gpurun_out/lt_${wl}_$lt.json 2> gpurun_out/lt_${wl}_$lt.err || { echo "bench $wl failed"; tail -5 gpurun_out/lt_${wl}_$lt.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/lt_${wl}_$lt.json').read().strip().splitlines()[-1]); print('$wl $lt', d['value'], d['ms_per_step'])"
done; done
