#!/bin/bash
# Re-collect only the PMC traffic tables (they are tied to the SHA-256 of iswm_amd/csrc: any source edit invalidates them and
# bench.py then reports roofline.traffic = null).  Run on the GPU box from the repo root:  tools/refresh_pmc.sh r03
tag=${1:-r03}
out=gpurun_out/profiles_$tag
mkdir -p $out/pmc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PMC_SEQ=$out/pmc/seq.json
rocprofv3 --pmc FETCH_SIZE -d $out/pmc/f -o f --output-format csv -- python3 tools/pmc_step.py > $out/pmc/f.log 2>&1
unset PMC_SEQ
rocprofv3 --pmc WRITE_SIZE -d $out/pmc/w -o w --output-format csv -- python3 tools/pmc_step.py > $out/pmc/w.log 2>&1
python3 tools/pmc_aggregate.py $out/pmc/f/f_counter_collection.csv $out/pmc/w/w_counter_collection.csv $out/pmc/step_traffic.json > $out/pmc/step_traffic_top.txt
python3 tools/pmc_by_geometry.py $out/pmc/f/f_counter_collection.csv $out/pmc/w/w_counter_collection.csv $out/pmc/seq.json > $out/pmc/traffic_by_geometry.txt 2>&1
rm -rf $out/pmc/f $out/pmc/w
head -3 $out/pmc/step_traffic_top.txt
