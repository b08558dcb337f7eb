#!/bin/bash
# A/B sweeps of the launch-plan knobs with the headline protocol (four images in flight), run on the GPU box:
#   bash tools/plan_ab.sh fc | wino | convsw | plan | traced | queues | soak          -> profiles/r06_fc_ranges.txt, r06_streams.txt
# Every line: images/s with four images in flight, then one image at a time (both on the plans the knobs select).
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
B="python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-events --no-resident --no-resnet --no-alt-math --no-repeats --no-latency-plan"
val() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d.get("images_per_s_other_protocols",{}); print(d["value"], o.get("one_image_at_a_time"))'; }
run() { echo "$2 $1: $(env $1 $B --math $2 2>/dev/null | val)"; }      # run "KNOB=v [KNOB=v]" <math>
case "$1" in
  fc)      # K ranges of the InnerProducts: FC_SPLIT_DIV digits = divisor for K > 50000 | K <= 8192 | other (0 = the full cut everywhere)
    for m in f16 mixed bf16x3; do for d in 0 2 4 8 42 82 84; do run MNC_FC_SPLIT_DIV=$d $m; done; done
    for d in 111 211 121 221 2; do run MNC_FC_SPLIT_DIV=$d fp32; done ;;
  wino)    # F(4x4) layers smaller than one round: slots they are cut to fill; tail cuts of the larger layers
    for f in 512 448 384 320 256 128; do run MNC_WINO_FILL=$f fp32; done
    run MNC_WINO_TAIL=0 fp32; run MNC_WINO_XCD=0 fp32 ;;
  convsw)  # conv_sw.hip: plan 0 from P0MIN workgroups on, plan 1 from P1MIN
    for m in f16 mixed; do for p in "384 128" "256 128" "256 64" "384 64" "64 64"; do set -- $p; run "MNC_CONVX3_P0MIN=$1 MNC_CONVX3_P1MIN=$2" $m; done; done ;;
  plan)    # the two plans as a whole
    for m in fp32 mixed f16 bf16 bf16x3; do run MNC_PLAN=0 $m; run MNC_PLAN=1 $m; done ;;
  traced)  # rocprofv3 kernel trace of both plans, what runs beside what (tools/stream_report.py)
    R=$PWD; cd /tmp && export TMPDIR=/tmp
    for m in fp32 f16; do for plan in 0 1; do
      rm -rf /tmp/tr
      MNC_PLAN=$plan timeout 600 rocprofv3 --kernel-trace -d /tmp/tr -o b -- python $R/bench.py --steps 160 --warmup 10 --math $m --no-cpu-baseline --no-events --no-resident --no-resnet --no-alt-math --no-repeats --no-latency-plan > /tmp/tr.log 2>&1
      echo "=== $m, MNC_PLAN=$plan, four images in flight"
      python $R/tools/stream_report.py /tmp/tr/b_results.db --images 30 130 | grep -v "columns of"
    done; done ;;
  queues)  # hardware queues x images in flight (the library's own default is 16 queues; an exported value wins)
    for q in 4 8 16; do for n in 4 8 12 16; do if [ $n -le $q ] || [ $q = 4 ]; then echo "fp32 GPU_MAX_HW_QUEUES=$q in-flight $n: $(GPU_MAX_HW_QUEUES=$q $B --math fp32 --in-flight $n 2>/dev/null | val)"; fi; done; done
    for m in f16 mixed; do for n in 4 12; do echo "$m in-flight $n: $($B --math $m --in-flight $n 2>/dev/null | val)"; done; done ;;
  soak)
    for p in "fp32 5000" "mixed 8000" "f16 10000"; do set -- $p; echo "$1 $2 steps: $(python bench.py --steps $2 --warmup 10 --math $1 --no-cpu-baseline --no-events --no-resident --no-resnet --no-alt-math --no-repeats --no-latency-plan 2>/dev/null | val)"; done ;;
  *) echo "usage: $0 fc | wino | convsw | plan | traced | queues | soak"; exit 2 ;;
esac
