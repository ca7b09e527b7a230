#!/bin/bash
# needs the ablation build: make -C ray-tracer_amd/csrc ablation (the release library ignores $KKD_DEBUG).  One-lane stage times per bit set.
for d in "$@"; do
  KKD_LIB=$PWD/ray-tracer_amd/lib_ab/libkkdirect.so KKD_DEBUG=$d timeout -k 10 300 python3 -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($d, j['value'], j['roofline']['stage_ms_per_step'])" || exit 1
done
