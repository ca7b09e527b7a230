#!/bin/bash
# A/B of a developer environment switch on one box with the default build: scripts/ab_envvar.sh VAR=value [bench args]
# (prints the bench line without and with the variable, twice)
kv="$1"; shift
for rep in 1 2; do
for mode in off on; do
  if [ "$mode" = on ]; then export "$kv"; else unset "${kv%%=*}"; fi
  timeout -k 10 200 python -u bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']
print('$kv $mode', j['value'], j['ms_per_step'], r['stage_ms_per_step'])" || exit 1
done
done
