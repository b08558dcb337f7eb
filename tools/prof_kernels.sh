#!/bin/bash
# PMC stall breakdown of kernels run by tools/kernel_bench.py (run on the GPU box).
# usage: bash tools/prof_kernels.sh <sql-like pattern, e.g. %conv3x3_sw%> <kernel_bench args...>
PAT=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_k
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/kernel_bench.py $* --reps 5"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $OUT/p1 -o k -- $CMD > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC -d $OUT/p2 -o k -- $CMD > $OUT/p2.log 2>&1
python - <<PY
import sqlite3
for p in ("p1","p2"):
    con=sqlite3.connect("$OUT/%s/k_results.db"%p)
    q="select kernel_name, grid_size, counter_name, avg(value), count(*) , avg(duration) from counters_collection where kernel_name like '$PAT' group by kernel_name, grid_size, counter_name order by grid_size, kernel_name"
    for r in con.execute(q):
        print("%-44s grid=%-9d %-30s avg=%-14.6g n=%d dur_us=%.1f"%(r[0].replace('void mnc::','')[:44],r[1],r[2],r[3],r[4],r[5]/1e3))
PY
