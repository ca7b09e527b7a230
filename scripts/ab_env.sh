#!/bin/bash
# A/B of environment settings on one box: scripts/ab_env.sh "VAR=1" "VAR=2 OTHER=3" ... ("-" = no setting); AB_ARGS = extra bench arguments
for rep in 1 2; do
for setting in "$@"; do
  if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
  env $envs timeout -k 10 300 python -u bench.py --steps 3 --warmup 2 --no-cpu-baseline ${AB_ARGS} 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']
print('[$setting]', j['value'], j['ms_per_step'], r['stage_ms_per_step'])" || exit 1
done
done
