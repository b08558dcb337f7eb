#!/bin/bash
# Variant builds of csrc/conv_sw.hip ON THE GPU BOX with extra -D flags, timed with tools/kernel_bench.py convsw (product semantics:
# results stay correct).   usage: tools/sw_var.sh "<flag sets, ';' separated>" "<modes>" [kernel_bench args]
cd "$(dirname "$0")/.."
IFS=';' read -ra sets <<< "$1"; modes=$2; shift 2
mkdir -p _abl
i=0
for fl in "${sets[@]}"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function $fl \
    -I mnc_amd/csrc -c mnc_amd/csrc/conv_sw.hip -o _abl/conv_sw_v$i.o || exit 1
  objs=$(ls mnc_amd/csrc/_obj/*.o | grep -v "/conv_sw.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _abl/libmnc_swv$i.so $objs _abl/conv_sw_v$i.o -ldl || exit 1
  for m in $modes; do
    echo "== convsw $m [$fl]"
    MNC_LIB_PATH=$PWD/_abl/libmnc_swv$i.so timeout 300 python tools/kernel_bench.py convsw --mode $m "$@" | awk '{printf "%s %s  ", $1, $5} END {print ""}'
  done
done
