#!/bin/bash
# A/B loop on the GPU box: a parity subset, then the bench of the named configurations for every library build listed in
# LIBS (space-separated tags of ice_halo_sim_amd/libhalo_hip_<tag>.so; "-" = the product build), all in ONE call so that
# the builds are compared on the same GPU.
#   gpurun -- 'LIBS="- a b" tools/quick_ab.sh 1 2'        (QUICK_TESTS=0 skips the parity subset)
CFGS=${@:-1 2}
for tag in ${LIBS:--}; do
  if [ "$tag" = "-" ]; then unset HALO_LIB; else export HALO_LIB=$(pwd)/ice_halo_sim_amd/libhalo_hip_$tag.so; fi
  echo "== build: $tag"
  if [ "${QUICK_TESTS:-1}" != "0" ]; then
    python -m pytest tests/test_gpu_parity.py tests/test_gpu_production_routes.py -m gpu -x -q -k "${QUICK_K:-lenses or config2 or full_size or frozen or multi_scatter_parity or chunking or config3}" 2>&1 | tail -3
  fi
  for c in $CFGS; do
    python bench.py --config $c --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; m=d.get('multi_scatter',{})
print('cfg$c value %.4e rays/s  ms/step %.2f  cov %.4f  dominant launch %.3f ms x %d  frac %.4f  first-layer %.3f ms' % (d['value'], d['ms_per_step'], d['repeats']['cov'], r['avg_launch_ms'], r['launches'], r['frac'], m.get('first_layer_kernel_ms_per_launch',0)))"
  done
done
