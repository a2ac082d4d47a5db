#!/bin/bash
# memory-side fetch (FETCH_SIZE, KiB; x2 on gfx950) of the ASPP weight gradients with and without the column-block-per-XCD order
#   tools/pmc_rect_xcd.sh <outfile>      (run on the GPU box)
out=${1:-gpurun_out/pmc_rect_xcd.txt}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f $out
for v in 0 1; do
  export ISWM_WG_RECT_XCD=$v
  for k in "2048 256 3 6 6" "2048 256 3 12 12" "2048 256 3 18 18"; do
    echo "== ISWM_WG_RECT_XCD=$v wgrad $k" >> $out
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmcx/a -o s --output-format csv -- python3 tools/pmc_wgrad.py $k > /dev/null 2>&1
    python3 tools/pmc_kernel_summary.py gpurun_out/pmcx/a/s_counter_collection.csv k_wgrad_pl >> $out
  done
done
rm -rf gpurun_out/pmcx
cat $out
