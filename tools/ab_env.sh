#!/bin/bash
# A/B an environment switch on ONE box: usage ab_env.sh VAR   (runs bench with VAR=0 and VAR=1, twice each)
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  env $1=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt --steps 20 > gpurun_out/ab_0_$i.log 2>/dev/null
  env $1=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-alt --steps 20 > gpurun_out/ab_1_$i.log 2>/dev/null
done
echo done
