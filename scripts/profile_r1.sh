#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + three separate PMC passes of the bench workload.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
# (a) the default command: two batch lanes in flight in the timed region, then bench.py's own one-lane pass
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.err || { echo trace failed; exit 1; }
# (b) the same command with every render on one lane: all kernel durations exclusive (what roofline.avg_launch_ms is)
KKD_STREAMS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_trace1.json 2> $OUT/trace1.err || { echo trace1 failed; exit 1; }
echo trace done
KKD_STREAMS=1 timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --spp 32 --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || { echo fetch failed; exit 1; }
echo fetch done
KKD_STREAMS=1 timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --spp 32 --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.err || { echo write failed; exit 1; }
echo write done
KKD_STREAMS=1 timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 bench.py --steps 1 --warmup 0 --spp 32 --no-cpu-baseline > $OUT/pmc_l2.json 2> $OUT/pmc_l2.err || { echo l2 failed; exit 1; }
echo l2 done
find $OUT -name "*.csv" | xargs ls -la | awk '{print $5, $9}'
