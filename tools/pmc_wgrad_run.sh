#!/bin/bash
# L2 hit / miss, memory-side bytes and SQ wait counters of the planes weight-gradient kernels on two shapes (run on the GPU box):
#   tools/pmc_wgrad_run.sh <outfile>
out=${1:-gpurun_out/pmc_wgrad.txt}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f $out
for k in "1024 256 1 0 1" "256 256 3 1 1" "512 512 3 2 2"; do
  echo "== wgrad $k" >> $out
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d gpurun_out/pmcw/a -o s --output-format csv -- python3 tools/pmc_wgrad.py $k > /dev/null 2>&1
  python3 tools/pmc_kernel_summary.py gpurun_out/pmcw/a/s_counter_collection.csv k_wgrad_pl >> $out
  rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmcw/b -o s --output-format csv -- python3 tools/pmc_wgrad.py $k > /dev/null 2>&1
  python3 tools/pmc_kernel_summary.py gpurun_out/pmcw/b/s_counter_collection.csv k_wgrad_pl >> $out
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d gpurun_out/pmcw/c -o s --output-format csv -- python3 tools/pmc_wgrad.py $k > /dev/null 2>&1
  python3 tools/pmc_kernel_summary.py gpurun_out/pmcw/c/s_counter_collection.csv k_wgrad_pl >> $out
done
rm -rf gpurun_out/pmcw
cat $out
