cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
export NCCL_DEBUG=INFO
for m in a b c; do
  echo "=== rccl_probe $m" 
  timeout 120 python tools/rccl_probe.py $m > $OUT/rccl_probe_$m.log 2>&1; echo "rc=$?"; grep -v "^\s*$" $OUT/rccl_probe_$m.log | tail -25
done
unset NCCL_DEBUG
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q > $OUT/pytest_pipeline.log 2>&1; echo "pipeline rc=$?"; tail -30 $OUT/pytest_pipeline.log
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_engine.py::test_device_instance_block_and_rccl_gather --ignore tests/test_gpu_pipeline.py > $OUT/pytest_gpu_r02b.log 2>&1; echo "all rc=$?"; tail -15 $OUT/pytest_gpu_r02b.log
