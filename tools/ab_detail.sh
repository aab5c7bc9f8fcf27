#!/bin/bash
# same-box A/B with the launch group's pieces: LIBS="tagA - tagB" tools/ab_detail.sh <cfg> [bench args]   ("-" = the product build)
CFG=${1:-4}; shift
for tag in ${LIBS:--}; do
  if [ "$tag" = "-" ]; then unset HALO_LIB; else export HALO_LIB=$(pwd)/ice_halo_sim_amd/libhalo_hip_$tag.so; fi
  python bench.py --config $CFG --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-8s cfg$CFG %.4e rays/s  ms/step %.3f  cov %.4f  trace %.3f  passes %.3f  group %.3f  wall/launch %.3f' % ('$tag', d['value'], d['ms_per_step'], d['repeats']['cov'], r['avg_launch_ms'], r['passes_ms_per_launch'], r['avg_launch_group_ms'], r['wall_ms_per_launch']))"
done
