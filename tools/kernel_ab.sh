#!/bin/bash
# Per-kernel times (rocprofv3 --kernel-trace --stats) of `bench.py --config CFG` for every library build in LIBS, in ONE call.
#   gpurun -- 'LIBS="- a b" KERNELS="split|accumulate_range" tools/kernel_ab.sh 4'
CFG=${1:-1}
ROOT=$(pwd)
export TMPDIR=/tmp
for tag in ${LIBS:--}; do
  if [ "$tag" = "-" ]; then unset HALO_LIB; else export HALO_LIB=$ROOT/ice_halo_sim_amd/libhalo_hip_$tag.so; fi
  out=$ROOT/gpurun_out/kab_$tag
  rm -rf $out
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $out -o k -- python $ROOT/bench.py --config $CFG --no-cpu-baseline --repeats 1 ${BENCH_ARGS:-} > /dev/null 2>&1)
  echo "== build: $tag"
  python tools/rocpd_summary.py $(find $out -name "*.db" | head -1) | grep -E "${KERNELS:-halo}" | cut -c1-60,90-170
  rm -rf $out
done
