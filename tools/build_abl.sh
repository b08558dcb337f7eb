#!/bin/bash
# tuning build of the F(4x4) kernel only: _abl/libmnc_f4abl.so = the product objects + conv_wino4.hip compiled with -DMNC_TUNING
# (kernel_bench convwino4 with MNC_LIB_PATH=_abl/libmnc_f4abl.so MNC_WINO_F4=<ablation bits>)
set -e
cd "$(dirname "$0")/.."
python -m mnc_amd._build > /dev/null
mkdir -p _abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DMNC_TUNING \
  -fno-slp-vectorize -I mnc_amd/csrc -c mnc_amd/csrc/conv_wino4.hip -o _abl/conv_wino4.o
objs=$(ls mnc_amd/csrc/_obj/*.o | grep -v conv_wino4.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _abl/libmnc_f4abl.so $objs _abl/conv_wino4.o -ldl
ls -la _abl/libmnc_f4abl.so
