#!/bin/bash
# args: "rounds capClosest capAny capLater"
for cfg in "$@"; do
  set -- $cfg
  KKD_TRACE_ROUNDS=$1 KKD_ITER_CAP_CLOSEST=$2 KKD_ITER_CAP_ANY=$3 KKD_ITER_CAP_LATER=$4 timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('cfg $cfg', j['value'], j['roofline']['stage_ms_per_step'])" || exit 1
done
