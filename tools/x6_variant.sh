#!/bin/bash
# timing experiment: rebuild conv_mfma_x6.hip with extra -D flags and time the forward kernels (speed section of x6_check.py)
cd "$(dirname "$0")/.."
i=0
for flags in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc $flags -c iswm_amd/csrc/conv_mfma_x6.hip -o /tmp/x6var.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o iswm_amd/libiswm_hip.so /tmp/x6var.o $(ls iswm_amd/build/*.o | grep -v "conv_mfma_x6.o") || exit 1
  echo "== $flags =="
  python tools/x6_check.py fwd > gpurun_out/x6var_$i.log 2>&1
  grep "fwd\[x6\]" gpurun_out/x6var_$i.log | sed 's/fwd\[f32\] *[0-9.]* us *[0-9.]* TF//'
done
