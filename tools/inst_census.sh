#!/bin/bash
# Runs on the GPU box: the trace kernel's dynamic instruction census by class as max_hits grows (slope = one interaction) and with every exit
# culled (difference = the exit queue's pops: projection + accumulation).  PMC passes only (+ --kernel-trace).
#   gpurun --timeout 900 -- bash tools/inst_census.sh   -> gpurun_out/inst_census.txt
set -u
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out/census; mkdir -p $OUT; cd /tmp
RES=$ROOT/gpurun_out/inst_census.txt; : > $RES
for V in normal away; do for H in 1 2 4 7; do
  T=h${H}_$V
  CENSUS_MAX_HITS=$H CENSUS_VIEW=$V timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $OUT/$T -o $T -- python $ROOT/tools/inst_census_launch.py > $OUT/$T.log 2>&1
  CENSUS_MAX_HITS=$H CENSUS_VIEW=$V timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES -d $OUT/${T}_b -o $T -- python $ROOT/tools/inst_census_launch.py > $OUT/${T}_b.log 2>&1
  echo "== max_hits $H view $V: $(tail -1 $OUT/$T.log)" >> $RES
  for d in $OUT/$T $OUT/${T}_b; do db=$(find $d -name "*.db" | head -1); [ -n "$db" ] && python $ROOT/tools/rocpd_summary.py $db | grep "halo_trace_kernel" >> $RES; rm -rf $d; done
done; done
cat $RES | cut -c1-200
