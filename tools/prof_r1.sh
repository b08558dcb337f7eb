#!/bin/bash
# Round-1 profiling recipe (run on the GPU box via gpurun).  Writes under gpurun_out/; summaries are copied to profiles/.
set -x
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 -L 2>/dev/null | grep -iE "mfma|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|GRBM_GUI_ACTIVE|SQ_INSTS_VALU |LDS_BANK_CONFLICT" | head -40 > $OUT/counters.txt
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT -d $OUT/pmc_mfma -o bench -- $BENCH > $OUT/pmc_mfma.log 2>&1
find $OUT -type f | head -50
ls -la $OUT/trace/* | head
