#!/bin/bash
# One gpurun call: the round's rocprofv3 summaries for every bench configuration that is priced in DESIGN.md.
#   gpurun --timeout 3000 -- bash tools/collect_all_profiles.sh r04
# then: cp gpurun_out/prof_<round>_*/<round>_*.txt profiles/
ROUND=${1:-r03}
for c in 1 2 4 4p; do bash tools/collect_profiles.sh $ROUND $c > gpurun_out/collect_${ROUND}_$c.log 2>&1; done
bash tools/collect_profiles.sh $ROUND filter:direction_out 0 ${ROUND}_filter_direction_out > gpurun_out/collect_${ROUND}_fd.log 2>&1
bash tools/collect_profiles.sh $ROUND filter:complex_PBD 0 ${ROUND}_filter_complex_PBD > gpurun_out/collect_${ROUND}_fc.log 2>&1
bash tools/collect_profiles.sh $ROUND filter:crystal_all_pass 0 ${ROUND}_filter_crystal_all_pass > gpurun_out/collect_${ROUND}_fa.log 2>&1
bash tools/collect_profiles.sh $ROUND ref:ms_multi_crystal_complex_filter 0 ${ROUND}_ref_ms_multi_crystal_complex_filter > gpurun_out/collect_${ROUND}_rc.log 2>&1
bash tools/collect_profiles.sh $ROUND ref:bench_light_single_ms 0 ${ROUND}_ref_bench_light_single_ms > gpurun_out/collect_${ROUND}_rl.log 2>&1
bash tools/collect_profiles.sh $ROUND ref:ms_multi_crystal 0 ${ROUND}_ref_ms_multi_crystal > gpurun_out/collect_${ROUND}_rm.log 2>&1
ls gpurun_out/prof_${ROUND}_*/*.txt | wc -l
