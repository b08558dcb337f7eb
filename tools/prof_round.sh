#!/bin/bash
# Per-round profiling recipe (run on the GPU box via gpurun): bench JSON + rocprofv3 kernel trace + 3 PMC passes.
# usage: bash tools/prof_round.sh <tag> [bench args]     -> gpurun_out/prof_<tag>/
TAG=${1:-rXX}
shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-events --no-resident --no-resnet --no-latency-plan $*"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
PMCB="$BENCH --no-graph --in-flight 1"      # counter passes: direct launches, one image at a time (same kernels in a fixed order per image; counters serialise the dispatches anyway)
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- $PMCB > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- $PMCB > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT -d $OUT/pmc_mfma -o bench -- $PMCB > $OUT/pmc_mfma.log 2>&1
python $REPO/tools/rocpd_summary.py $OUT/trace/bench_results.db > $OUT/kernel_trace_stats.txt
# one image at a time, fp32 only, every step on the captured graph: per-kernel table of the headline mode + where the stream idles
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace1 -o bench -- python $REPO/bench.py --steps 12 --warmup 4 --in-flight 1 --no-cpu-baseline --no-events --no-resident --no-resnet --no-alt-math --no-latency-plan > $OUT/trace1.log 2>&1
python $REPO/tools/rocpd_summary.py $OUT/trace1/bench_results.db > $OUT/kernel_trace_stats_fp32_serial.txt
python $REPO/tools/timeline_gaps.py $OUT/trace1/bench_results.db --top 15 > $OUT/timeline_gaps_fp32_serial.txt
rm -rf $OUT/trace1
BUILD=$(cd $REPO && python -m mnc_amd._build --hash)
python $REPO/tools/pmc_report.py $OUT/pmc_mfma/bench_results.db $OUT/pmc_fetch/bench_results.db $OUT/pmc_write/bench_results.db --json $OUT/pmc.json --build $BUILD --cycle=fc_mfma_dma=6 '--cycle=conv3x3_wino4_kernel<1, 0>=6' '--cycle=conv3x3_sw_kernel<2, 5, 2, 2, 1>=13' '--cycle=conv3x3_sw_kernel<1, 5, 2, 2, 1>=13' '--cycle=conv3x3_sw_kernel<0, 5, 2, 2, 1>=13' > $OUT/pmc.txt
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma
# the bench line LAST, with this build's counter profile in place: its roofline.traffic comes from profiles/pmc_latest.json and is
# reported only when that file carries this build's hash (copy $OUT/pmc.json to profiles/pmc_latest.json in the repo afterwards)
cp $OUT/pmc.json $REPO/profiles/pmc_latest.json
python $REPO/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
head -12 $OUT/kernel_trace_stats.txt
head -8 $OUT/pmc.txt
