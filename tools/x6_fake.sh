#!/bin/bash
# timing experiment: how fast would the bf16x6 forward be if the operand split cost nothing?
set -e
cd "$(dirname "$0")/.."
for lvl in 1 2; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DISWM_X6_FAKE=$lvl -c iswm_amd/csrc/conv_mfma_x6.hip -o /tmp/x6fake.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o iswm_amd/libiswm_hip.so /tmp/x6fake.o $(ls iswm_amd/build/*.o | grep -v conv_mfma_x6)
  echo "== FAKE level $lvl (1: B split free, 2: A and B split free) =="
  python tools/x6_check.py fwd 2>&1 | grep "fwd\["
done
