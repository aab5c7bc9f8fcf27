#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of `bench.py --config CFG`, then PMC passes in their own runs
# (counters are never collected together with trace domains other than --kernel-trace).
# Output: gpurun_out/prof_<tag>/<round>_bench<cfg>_*.txt — copy the summaries into profiles/.
#   tools/collect_profiles.sh <round> <cfg> [rays-per-session] [tag]      (cfg may also be filter:<case>)
# cfg: 1 | 2 | 4 | 4d | 4p | ref:<document>; tag names the output files (default <round>_bench<cfg>, ':' dropped)
# LITE=1: the kernel table and the FETCH / WRITE_SIZE pair only (what bench.py reads for the reference's documents)
set -u
ROUND=${1:-r03}
CFG=${2:-1}
RAYS=${3:-0}
TAG=${4:-${ROUND}_bench${CFG//:/_}}
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--config $CFG --no-cpu-baseline --no-others --repeats 1 --rays-per-wl $RAYS"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o $TAG -- python $ROOT/bench.py --steps 2 --warmup 1 $ARGS > $OUT/stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -o $TAG -- python $ROOT/bench.py --steps 1 --warmup 0 $ARGS > $OUT/pmc_$C.log 2>&1
done
if [ -z "${LITE:-}" ]; then
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $OUT/pmc_insts -o $TAG -- python $ROOT/bench.py --steps 1 --warmup 0 $ARGS > $OUT/pmc_insts.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -d $OUT/pmc_cycles -o $TAG -- python $ROOT/bench.py --steps 1 --warmup 0 $ARGS > $OUT/pmc_cycles.log 2>&1
# dynamic instruction census by class (what the issue-fraction estimate in bench.py prices with tools/valu_rate_bench's costs) and the
# wave-cycle decomposition (busy = active + waiting for an instruction to issue + parked at s_waitcnt / barrier)
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $OUT/pmc_classes -o $TAG -- python $ROOT/bench.py --steps 1 --warmup 0 $ARGS > $OUT/pmc_classes.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc_wait -o $TAG -- python $ROOT/bench.py --steps 1 --warmup 0 $ARGS > $OUT/pmc_wait.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU -d $OUT/pmc_lds -o $TAG -- python $ROOT/bench.py --steps 1 --warmup 0 $ARGS > $OUT/pmc_lds.log 2>&1
fi
cd $ROOT
name() { case $1 in stats) echo kernel_stats;; pmc_FETCH_SIZE) echo pmc_fetch_size;; pmc_WRITE_SIZE) echo pmc_write_size;; *) echo $1;; esac; }
for d in stats pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_insts pmc_cycles pmc_classes pmc_wait pmc_lds; do
  db=$(find $OUT/$d -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db > $OUT/${TAG}_$(name $d).txt 2>&1
  rm -rf $OUT/$d
done
tail -2 $OUT/stats.log | cut -c1-400
ls -la $OUT
