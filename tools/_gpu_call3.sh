cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_engine.py::test_device_instance_block_and_rccl_gather -x -q > $OUT/pytest_pipeline.log 2>&1; echo "pipeline rc=$?"; tail -30 $OUT/pytest_pipeline.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29617 \
  bench.py --gpus 1 --steps 100 --warmup 5 > $OUT/bench_launcher1_r02c.json 2> $OUT/bench_launcher1_r02c.err
echo "launcher bench rc=$?"; cut -c1-1500 $OUT/bench_launcher1_r02c.json; grep -v "^\s*$" $OUT/bench_launcher1_r02c.err | tail -8
timeout 900 python bench.py > $OUT/bench_r02c.json 2> $OUT/bench_r02c.err; echo "bench rc=$?"; cat $OUT/bench_r02c.json; tail -5 $OUT/bench_r02c.err
