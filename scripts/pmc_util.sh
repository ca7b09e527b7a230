#!/bin/bash
# SQ / TA / TCP utilisation counters per kernel stage (two separate passes), low-spp C2 render.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_util; rm -rf $OUT; mkdir -p $OUT
KKD_STREAMS=1 timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $OUT/sq -- python3 bench.py --steps 1 --warmup 0 --spp 32 --no-cpu-baseline > $OUT/sq.json 2> $OUT/sq.err || { echo sq failed; tail -3 $OUT/sq.err; exit 1; }
KKD_STREAMS=1 timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/ta -- python3 bench.py --steps 1 --warmup 0 --spp 32 --no-cpu-baseline > $OUT/ta.json 2> $OUT/ta.err || { echo ta failed; tail -3 $OUT/ta.err; exit 1; }
echo ok
