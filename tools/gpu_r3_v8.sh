#!/bin/bash
# round 3, GPU visit 8: per-kernel durations of the C5 decode step with the consumer-prologue merges (rocprofv3 kernel trace)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
for MODE in "1 1" "1 0"; do
  set -- $MODE
  TAG=kv$1_mlp$2
  (cd /tmp && GM_DECODE_KV_FUSE=$1 GM_DECODE_MLP_FUSE=$2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/r3v8_prof_$TAG -o c5 -- python $R/tools/diag_c5.py > $R/$OUT/r3v8_prof_$TAG.log 2>&1)
  F=$(find $OUT/r3v8_prof_$TAG -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $OUT/r3v8_c5_kernel_stats_$TAG.csv
  rm -rf $OUT/r3v8_prof_$TAG
done
