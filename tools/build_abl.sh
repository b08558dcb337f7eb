#!/bin/bash
# tuning build of ONE kernel file: _abl/libmnc_<name>abl.so = the product objects + csrc/<file>.hip compiled with -DMNC_TUNING
#   tools/build_abl.sh                 conv_wino4.hip -> _abl/libmnc_f4abl.so  (kernel_bench convwino4, MNC_WINO_F4=<ablation bits>)
#   tools/build_abl.sh gemm fc         gemm.hip       -> _abl/libmnc_fcabl.so  (kernel_bench fc, MNC_FC_DMA_ABL=16 + bits)
# used as MNC_LIB_PATH=_abl/libmnc_...so
set -e
cd "$(dirname "$0")/.."
src=${1:-conv_wino4}; tag=${2:-f4}
python -m mnc_amd._build > /dev/null
mkdir -p _abl
extra=""; [ "$src" = conv_wino4 ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DMNC_TUNING \
  $extra -I mnc_amd/csrc -c mnc_amd/csrc/$src.hip -o _abl/$src.o
objs=$(ls mnc_amd/csrc/_obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _abl/libmnc_${tag}abl.so $objs _abl/$src.o -ldl
ls -la _abl/libmnc_${tag}abl.so
