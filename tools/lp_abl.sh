#!/bin/bash
# Variant builds of csrc/gemm_x3.hip ON THE GPU BOX with extra -D flags (MNC_LP_ABL ablations: wrong results, timing only),
# timed with tools/kernel_bench.py.   usage: tools/lp_abl.sh "<flag sets, ';' separated>" <kernel_bench args ...>
cd "$(dirname "$0")/.."
IFS=';' read -ra sets <<< "$1"; shift
mkdir -p _abl
i=0
for fl in "${sets[@]}"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function $fl \
    -I mnc_amd/csrc -c mnc_amd/csrc/gemm_x3.hip -o _abl/gemm_x3_v$i.o || exit 1
  objs=$(ls mnc_amd/csrc/_obj/*.o | grep -v "/gemm_x3.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _abl/libmnc_lpv$i.so $objs _abl/gemm_x3_v$i.o -ldl || exit 1
  echo "== [$fl] $*"
  MNC_LIB_PATH=$PWD/_abl/libmnc_lpv$i.so timeout 300 python tools/kernel_bench.py "$@"
done
