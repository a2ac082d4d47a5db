#!/bin/bash
# Regenerates the measurement artefacts kept under profiles/ (run on the GPU box from the repo root):
#   tools/collect_profiles.sh r03
tag=${1:-r03}
out=gpurun_out/profiles_$tag
mkdir -p $out/pmc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 --torch-baseline > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python3 bench.py --steps 10 --warmup 3 --no-alt --no-cpu-baseline --no-kernel-timing > $out/bench_rocprof.json 2> $out/bench_rocprof.err
python3 tools/prof_summary.py $out/kt/kt_results.db 14 $out/bench_kernel_stats.txt > /dev/null
python3 tools/prof_sequence.py $out/kt/kt_results.db 2 $out/step_sequence.txt > /dev/null
python3 tools/conv_table.py resnet101 16 > $out/conv_table.txt 2>&1
export PMC_SEQ=$out/pmc/seq.json
rocprofv3 --pmc FETCH_SIZE -d $out/pmc/f -o f --output-format csv -- python3 tools/pmc_step.py > $out/pmc/f.log 2>&1
unset PMC_SEQ
rocprofv3 --pmc WRITE_SIZE -d $out/pmc/w -o w --output-format csv -- python3 tools/pmc_step.py > $out/pmc/w.log 2>&1
python3 tools/pmc_aggregate.py $out/pmc/f/f_counter_collection.csv $out/pmc/w/w_counter_collection.csv $out/pmc/step_traffic.json > $out/pmc/step_traffic_top.txt
python3 tools/pmc_by_geometry.py $out/pmc/f/f_counter_collection.csv $out/pmc/w/w_counter_collection.csv $out/pmc/seq.json > $out/pmc/traffic_by_geometry.txt 2>&1
python3 tools/pl2_shapes.py 20 > $out/pl2_shapes.txt 2>&1
python3 tools/ew_bench.py > $out/elementwise.txt 2>&1
python3 tools/acc_check.py > $out/acc_vs_fp64.txt 2>&1
rm -rf $out/kt $out/pmc/f $out/pmc/w
ls -la $out $out/pmc
