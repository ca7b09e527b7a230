#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + stats of the bench command, two lanes (the timed configuration) and
# one lane (exclusive kernel durations, what roofline.avg_launch_ms is).   scripts/profile_trace.sh <workload> <outdir>
set -u
WL=${1:-c4}; OUT=${2:-gpurun_out/trace_$WL}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/two_lanes -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_two_lanes.json 2> $OUT/two_lanes.err || { echo trace failed; tail -3 $OUT/two_lanes.err; exit 1; }
KKD_STREAMS=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/one_lane -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_one_lane.json 2> $OUT/one_lane.err || { echo trace1 failed; tail -3 $OUT/one_lane.err; exit 1; }
find $OUT -name "*kernel_stats.csv" | xargs ls -la | awk '{print $5, $9}'
