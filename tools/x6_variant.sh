#!/bin/bash
# timing experiment: rebuild conv_mfma_x6.hip with extra -D flags and time the forward kernels
set -e
cd "$(dirname "$0")/.."
for flags in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc $flags -c iswm_amd/csrc/conv_mfma_x6.hip -o /tmp/x6var.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o iswm_amd/libiswm_hip.so /tmp/x6var.o $(ls iswm_amd/build/*.o | grep -v conv_mfma_x6)
  echo "== $flags =="
  python tools/x6_check.py fwd 2>&1 | grep "fwd\[" | sed 's/fwd\[f32\][^f]*//'
done
