#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the default bench, then PMC passes in their own runs.
# Output: gpurun_out/prof_<tag>/  (rocpd sqlite) — summarise with tools/rocpd_summary.py into profiles/.
set -u
TAG=${1:-r01b}
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o $TAG -- $BENCH > $OUT/stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -o $TAG -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$C.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $OUT/pmc_insts -o $TAG -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_insts.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -d $OUT/pmc_cycles -o $TAG -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_cycles.log 2>&1
cd $ROOT
for d in stats pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_insts pmc_cycles; do
  db=$(find $OUT/$d -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db > $OUT/$d.txt 2>&1
  find $OUT/$d -name "*.db" -size +30M -delete
done
tail -2 $OUT/stats.log
ls -la $OUT
