#!/bin/bash
# isolated per-shape A/B of two builds on ONE box: iswm_amd/build/libiswm_prev.so (A) vs iswm_amd/libiswm_hip.so (B)
#   tools/ab_shapes.sh [rounds] [geometry filter]   -> gpurun_out/ab_shapes_{A,B}.txt
cd "$GRAFT_REPO_ROOT"
cp iswm_amd/libiswm_hip.so /tmp/lib_new.so
cp iswm_amd/build/libiswm_prev.so iswm_amd/libiswm_hip.so
python tools/pl2_shapes.py ${1:-20} "$2" > gpurun_out/ab_shapes_A.txt 2>&1
cp /tmp/lib_new.so iswm_amd/libiswm_hip.so
python tools/pl2_shapes.py ${1:-20} "$2" > gpurun_out/ab_shapes_B.txt 2>&1
paste -d'\n' gpurun_out/ab_shapes_A.txt gpurun_out/ab_shapes_B.txt
