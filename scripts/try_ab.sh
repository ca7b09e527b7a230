#!/bin/bash
# On the GPU box: parity tests (+ test files named as arguments), then C4 / C2 throughput (no CPU baseline, no extra passes).
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_rounds.py "$@" -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pytest_ab.log 2>&1 || { tail -30 gpurun_out/pytest_ab.log; exit 1; }
tail -2 gpurun_out/pytest_ab.log
for wl in c4 c2; do
  timeout -k 10 300 python3 bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-extra-passes --no-secondary > gpurun_out/ab_${wl}.json 2> gpurun_out/ab_${wl}.err || { echo "bench $wl failed"; tail -5 gpurun_out/ab_${wl}.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/ab_${wl}.json').read().strip().splitlines()[-1]); print('$wl', d['value'], d['ms_per_step'])"
done
