#!/bin/bash
# Regenerates the measurement artefacts kept under profiles/ (run on the GPU box from the repo root):
#   tools/collect_profiles.sh r02
tag=${1:-r02}
out=gpurun_out/profiles_$tag
mkdir -p $out/pmc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python3 bench.py --steps 10 --warmup 3 --no-alt --no-cpu-baseline --no-kernel-timing > $out/bench_rocprof.json 2> $out/bench_rocprof.err
python3 tools/prof_summary.py $out/kt/kt_results.db 14 $out/bench_kernel_stats.txt > /dev/null
python3 tools/conv_table.py resnet101 16 > $out/conv_table.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/pmc/f -o f --output-format csv -- python3 tools/pmc_step.py > $out/pmc/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/pmc/w -o w --output-format csv -- python3 tools/pmc_step.py > $out/pmc/w.log 2>&1
python3 tools/pmc_aggregate.py $out/pmc/f/f_counter_collection.csv $out/pmc/w/w_counter_collection.csv $out/pmc/step_traffic.json > $out/pmc/step_traffic_top.txt
python3 tools/pl_check.py 20 > $out/pl_check.txt 2>&1
python3 tools/pl2_timeline.py > $out/pl2_timeline.txt 2>&1
python3 tools/ew_bench.py > $out/elementwise.txt 2>&1
python3 tools/wgrad_timeline.py 256 256 3 1 1 > $out/wgrad_timeline.txt 2>&1
python3 tools/pl2_ab.py 30 > $out/pl2_stamps_ab.txt 2>&1
hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o /tmp/mfma_rate > /dev/null 2>&1 && timeout -k 5 200 /tmp/mfma_rate > $out/mfma_rate.txt 2>&1
for k in "pl2 1024 256 1 0 1" "pl2 256 256 3 1 1" "wgrad 1024 256 1 0 1" "wgrad 256 256 3 1 1"; do
  set -- $k; kind=$1; shift
  name=$(echo $kind $* | tr ' ' '_')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d $out/pmc/sq1 -o s --output-format csv -- python3 tools/pmc_$kind.py $* > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $out/pmc/sq2 -o s --output-format csv -- python3 tools/pmc_$kind.py $* > /dev/null 2>&1
  pat=k_conv_pl2; [ $kind = wgrad ] && pat=k_wgrad_pl
  { echo "== $k"; python3 tools/pmc_kernel_summary.py $out/pmc/sq1/s_counter_collection.csv $pat; python3 tools/pmc_kernel_summary.py $out/pmc/sq2/s_counter_collection.csv $pat; } >> $out/pmc/sq_counters.txt
done
rm -rf $out/kt $out/pmc/sq1 $out/pmc/sq2
ls -la $out $out/pmc
