#!/bin/bash
# A/B an environment switch on ONE GPU box (boxes differ by up to ~10 %):  tools/ab_env_bench.sh VAR v1 v2 [v3 ...]
var=$1; shift
for round in 1 2; do
  for v in "$@"; do
    env $var=$v python bench.py --steps 10 --warmup 3 --no-alt --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', d['value'], 'img/s', d['ms_per_step'], 'ms/step')"
  done
done
