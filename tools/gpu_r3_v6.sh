#!/bin/bash
# round 3, GPU visit 6: C5 decode step with the attention-partial merge folded into the out-projection: parity + timing
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
LOG=$OUT/r3v6_round.log
echo "$(date)" > $LOG
timeout 900 python -m pytest tests/test_gpu_fullsize_oracle_r3.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_models.py -m gpu -q -k "c5 or transformer" --maxfail=10 -rP -p no:cacheprovider > $OUT/r3v6_tests.log 2>&1
echo "tests rc=$?" >> $LOG
grep -n "^_____.* test_\|^E  \|passed\|failed" $OUT/r3v6_tests.log | head -40 >> $LOG
grep "\[parity\]" $OUT/r3v6_tests.log > $OUT/r3v6_parity.txt
for MODE in "1" "0" "1"; do
  echo "kv_fuse=$MODE" >> $OUT/r3v6_diag.txt
  GM_DECODE_KV_FUSE=$MODE timeout 300 python tools/diag_c5.py >> $OUT/r3v6_diag.txt 2>&1
done
cat $OUT/r3v6_diag.txt >> $LOG
timeout 600 python tools/bench_c5.py > $OUT/r3v6_c5.json 2> $OUT/r3v6_c5.err; tail -c 1200 $OUT/r3v6_c5.json >> $LOG
echo "done $(date)" >> $LOG
