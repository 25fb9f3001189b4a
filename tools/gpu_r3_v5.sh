#!/bin/bash
# round 3, GPU visit 5: C5 decode step -- split-KV single-query attention + K-split small-row GEMMs: parity, A/B timing, kernel stats
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
LOG=$OUT/r3v5_round.log
echo "$(date)" > $LOG
timeout 900 python -m pytest tests/test_gpu_fullsize_oracle_r3.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_models.py -m gpu -q -k "c5 or transformer" --maxfail=10 -rP -p no:cacheprovider > $OUT/r3v5_tests.log 2>&1
echo "tests rc=$?" >> $LOG
grep -n "^_____.* test_\|^E  \|passed\|failed" $OUT/r3v5_tests.log | head -40 >> $LOG
grep "\[parity\]" $OUT/r3v5_tests.log > $OUT/r3v5_parity.txt
for MODE in "1 1" "0 1" "1 0" "0 0" "1 1"; do
  set -- $MODE
  echo "kv_split=$1 ksplit=$2" >> $OUT/r3v5_diag.txt
  GM_DECODE_KV_SPLIT=$1 GM_LINEAR_KSPLIT=$2 timeout 300 python tools/diag_c5.py >> $OUT/r3v5_diag.txt 2>&1
done
cat $OUT/r3v5_diag.txt >> $LOG
timeout 600 python tools/bench_c5.py > $OUT/r3v5_c5.json 2> $OUT/r3v5_c5.err; tail -c 1200 $OUT/r3v5_c5.json >> $LOG
R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/r3v5_prof -o c5 -- python $R/tools/diag_c5.py > $R/$OUT/r3v5_prof.log 2>&1)
F=$(find $OUT/r3v5_prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $OUT/r3v5_c5_kernel_stats.csv && head -14 $F | cut -c1-200 >> $LOG
rm -rf $OUT/r3v5_prof/*/*kernel_trace.csv
echo "done $(date)" >> $LOG
