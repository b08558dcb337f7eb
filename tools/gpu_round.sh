#!/bin/bash
# One gpurun call: GPU test suite, smoke, the bench line (plain and as 1 rank under the launcher), profiles.
# usage: bash tools/gpu_round.sh <tag> [--skip-tests] [--skip-prof]
TAG=${1:-rXX}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
rm -f $OUT/parity_report.txt
if [[ "$*" != *--skip-tests* ]]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_$TAG.log 2>&1
  echo "pytest rc=$?" | tee -a $OUT/pytest_gpu_$TAG.log
  tail -15 $OUT/pytest_gpu_$TAG.log
  timeout 300 python __graft_entry__.py --smoke > $OUT/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke_$TAG.log
fi
# 1 rank under the launcher: RCCL communicator + ncclAllGather of the device block inside every step
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29617 \
  bench.py --gpus 1 --steps 100 --warmup 5 > $OUT/bench_launcher1_$TAG.json 2> $OUT/bench_launcher1_$TAG.err
echo "launcher bench rc=$?"; cat $OUT/bench_launcher1_$TAG.json | cut -c1-600; tail -5 $OUT/bench_launcher1_$TAG.err
if [[ "$*" != *--skip-prof* ]]; then
  bash tools/prof_round.sh $TAG
else
  python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; cat $OUT/bench_$TAG.json
fi
