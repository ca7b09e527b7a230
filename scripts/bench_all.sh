#!/bin/bash
# One line per BASELINE config on one GPU (numbers for BASELINE.md / DESIGN.md); C5 = C4 at 4096 spp, here on ONE GPU.
mkdir -p gpurun_out
for wl in c1 c2 c3 c4; do
  timeout -k 10 500 python3 -u bench.py --workload $wl --steps 3 --warmup 2 --no-cpu-baseline --no-extra-passes --no-secondary > gpurun_out/all_$wl.json 2> gpurun_out/all_$wl.err || { echo "$wl failed"; tail -3 gpurun_out/all_$wl.err; exit 1; }
  python3 -c "
import json; j=json.loads(open('gpurun_out/all_$wl.json').read().strip().splitlines()[-1])
print('$wl', j['config']['workload'], j['value'], 'Msamples/s', j['mrays_per_s'], 'Mrays/s', j['ms_per_step'], 'ms', 'upload', j['config']['scene_upload_s'])"
done
timeout -k 10 500 python3 -u bench.py --workload c4 --spp 4096 --steps 1 --warmup 0 --no-cpu-baseline --no-extra-passes --no-secondary > gpurun_out/all_c5.json 2> gpurun_out/all_c5.err || { echo "c5 failed"; tail -3 gpurun_out/all_c5.err; exit 1; }
python3 -c "
import json; j=json.loads(open('gpurun_out/all_c5.json').read().strip().splitlines()[-1])
print('c5 (one GPU)', j['value'], 'Msamples/s', j['mrays_per_s'], 'Mrays/s', j['ms_per_step'], 'ms')"
