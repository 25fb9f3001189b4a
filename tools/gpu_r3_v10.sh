#!/bin/bash
# round 3, GPU visit 10: C5 decode step, merges in consumer prologues (attention partials, MLP K-slice partials): parity + timing
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
LOG=$OUT/r3v10_round.log
echo "$(date)" > $LOG
timeout 900 python -m pytest tests/test_gpu_fullsize_oracle_r3.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_models.py -m gpu -q -k "c5 or transformer" --maxfail=10 -rP -p no:cacheprovider > $OUT/r3v10_tests.log 2>&1
echo "tests rc=$?" >> $LOG
grep -n "^_____.* test_\|^E  \|passed\|failed" $OUT/r3v10_tests.log | head -40 >> $LOG
grep "\[parity\]" $OUT/r3v10_tests.log > $OUT/r3v10_parity.txt
for MODE in 1 0 1; do
  echo "attn_fast=$MODE" >> $OUT/r3v10_diag.txt
  GM_DECODE_ATTN_FAST=$MODE timeout 300 python tools/diag_c5.py >> $OUT/r3v10_diag.txt 2>&1
done
cat $OUT/r3v10_diag.txt >> $LOG
timeout 600 python tools/bench_c5.py > $OUT/r3v10_c5.json 2> $OUT/r3v10_c5.err; tail -c 1200 $OUT/r3v10_c5.json >> $LOG
timeout 600 python tools/bench_c5.py 4096 graph > $OUT/r3v10_c5_graph.json 2> $OUT/r3v10_c5_graph.err; tail -c 1200 $OUT/r3v10_c5_graph.json >> $LOG
echo "done $(date)" >> $LOG
R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/r3v10_prof -o c5 -- python $R/tools/diag_c5.py > $R/$OUT/r3v10_prof.log 2>&1)
F=$(find $OUT/r3v10_prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $OUT/r3v10_c5_kernel_stats.csv
rm -rf $OUT/r3v10_prof
