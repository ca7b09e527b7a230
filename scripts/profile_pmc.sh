#!/bin/bash
# Run on the GPU box (via gpurun): PMC passes of one bench workload, one lane, low spp (same batch size as the headline run).
#   scripts/profile_pmc.sh <workload> <spp> <outdir>
# Each pass is its own rocprofv3 run (counters are not split over passes); the program follows "--" directly.
set -u
WL=${1:-c4}; SPP=${2:-32}; OUT=${3:-gpurun_out/pmc_$WL}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $OUT; mkdir -p $OUT
export KKD_STREAMS=1
ARGS="bench.py --workload $WL --steps 1 --warmup 0 --spp $SPP --no-cpu-baseline --no-extra-passes"
pass() {  # name counters...
  local name=$1; shift
  timeout -s KILL 400 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ARGS > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -3 $OUT/$name.err; return 1; }
  echo "$name done"
}
pass fetch FETCH_SIZE && pass write WRITE_SIZE && pass l2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum \
 && pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU \
 && pass ta TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
find $OUT -name "*counter_collection.csv" | xargs ls -la | awk '{print $5, $9}'
