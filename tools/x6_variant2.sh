#!/bin/bash
# scratch ablation builds: compile SRC (a modified copy of conv_mfma_x6.hip) with each flag set, time with x6_ablate.py
cd "$(dirname "$0")/.."
SRC=$1; shift
for flags in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Iiswm_amd/csrc -Iinclude $flags -c $SRC -o /tmp/x6var.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o iswm_amd/libiswm_hip.so /tmp/x6var.o $(ls iswm_amd/build/*.o | grep -v "conv_mfma_x6.o") || exit 1
  echo "== $flags =="
  python tools/x6_ablate.py 2>&1 | grep " us"
done
