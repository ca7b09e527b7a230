#!/bin/bash
# A/B of experimental library builds on one box: scripts/ab.sh <libdir> [<libdir> ...]; "-" = the default build
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset KKD_LIB; else export KKD_LIB=$PWD/$lib/libkkdirect.so; fi
  timeout -k 10 200 python -u bench.py --steps 3 --warmup 2 --no-cpu-baseline ${AB_ARGS} 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']
print('$lib', j['value'], j['ms_per_step'], r['stage_ms_per_step'])" || exit 1
done
done
