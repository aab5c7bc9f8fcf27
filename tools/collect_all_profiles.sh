#!/bin/bash
# One gpurun call: the round's rocprofv3 summaries for every configuration bench.py prices (bench.profile_files_read() lists the files;
# tests/test_bench_profiles.py fails while one is missing).
#   gpurun --timeout 3000 -- bash tools/collect_all_profiles.sh r06
# then: cp gpurun_out/prof_<round>_*/<round>_*.txt profiles/
ROUND=${1:-r06}
for c in 1 2 4 4d 4p; do bash tools/collect_profiles.sh $ROUND $c > gpurun_out/collect_${ROUND}_$c.log 2>&1; done
for d in bench_light_single_ms ms_multi_crystal ms_multi_crystal_complex_filter ms_multi_crystal_filtered_bd config_example; do
  LITE=1 bash tools/collect_profiles.sh $ROUND ref:$d 0 ${ROUND}_ref_$d > gpurun_out/collect_${ROUND}_ref_$d.log 2>&1
done
if [ -n "${FILTERS:-}" ]; then
  for f in direction_out complex_PBD crystal_all_pass; do bash tools/collect_profiles.sh $ROUND filter:$f 0 ${ROUND}_filter_$f > gpurun_out/collect_${ROUND}_filter_$f.log 2>&1; done
fi
ls gpurun_out/prof_${ROUND}_*/*.txt | wc -l
