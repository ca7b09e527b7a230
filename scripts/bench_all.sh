#!/bin/bash
# One line per BASELINE config on one GPU (final numbers for BASELINE.md section 6)
for wl in c1 c2 c3 c4; do
  timeout -k 10 500 python -u bench.py --workload $wl --steps 2 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']
print('$wl', j['value'], j['mrays_per_s'], r['achieved'], r['frac'], j['ms_per_step'], r['stage_ms_per_step'])" || exit 1
done
