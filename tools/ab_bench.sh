#!/bin/bash
# A/B two builds of libiswm_hip.so on ONE box: iswm_amd/build/libiswm_prev.so (A) vs iswm_amd/libiswm_hip.so (B)
set -e
cd "$GRAFT_REPO_ROOT"
cp iswm_amd/libiswm_hip.so /tmp/lib_new.so
for i in 1 2; do
  cp iswm_amd/build/libiswm_prev.so iswm_amd/libiswm_hip.so
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt --steps 20 > gpurun_out/ab_A_$i.log 2>/dev/null
  cp /tmp/lib_new.so iswm_amd/libiswm_hip.so
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt --steps 20 > gpurun_out/ab_B_$i.log 2>/dev/null
done
echo done
