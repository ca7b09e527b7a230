#!/bin/bash
for b in "$@"; do
  timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print($b, j['value'], j['roofline']['stage_ms_per_step'])"
done
