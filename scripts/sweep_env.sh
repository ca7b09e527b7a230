#!/bin/bash
# usage: sweep_env.sh VAR v1 v2 ...
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$var=$v', j['value'], j['roofline']['stage_ms_per_step'])" || exit 1
done
