#!/bin/bash
# Ablation builds of csrc/conv_sw.hip ON THE GPU BOX (timing only, wrong results): _abl/libmnc_swabl<N>.so = the product objects +
# conv_sw.hip compiled with -DMNC_SW_ABL=<N> (1 no copies in the loop, 2 no fragment reads, 4 no output stores); then
# tools/kernel_bench.py convsw under each.   usage: tools/sw_abl.sh "<abl values>" "<modes>" [kernel_bench args]
cd "$(dirname "$0")/.."
abls=${1:-"1 2 4 7"}; modes=${2:-"f16 bf16x3"}; shift 2
mkdir -p _abl
for a in $abls; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DMNC_SW_ABL=$a \
    -I mnc_amd/csrc -c mnc_amd/csrc/conv_sw.hip -o _abl/conv_sw_$a.o || exit 1
  objs=$(ls mnc_amd/csrc/_obj/*.o | grep -v "/conv_sw.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _abl/libmnc_swabl$a.so $objs _abl/conv_sw_$a.o -ldl || exit 1
  for m in $modes; do
    echo "== convsw $m ABL $a"
    MNC_LIB_PATH=$PWD/_abl/libmnc_swabl$a.so timeout 300 python tools/kernel_bench.py convsw --mode $m "$@"
  done
done
