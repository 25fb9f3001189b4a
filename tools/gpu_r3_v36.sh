#!/bin/bash
# round 3, GPU visit 36: C5 decode step with LayerNorm + q|k|v + attention ranges as one launch: parity + A/B
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
LOG=$OUT/r3v36_round.log
echo "$(date)" > $LOG
timeout 900 python -m pytest tests/test_gpu_fullsize_oracle_r3.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_models.py -m gpu -q -k "c5 or transformer" --maxfail=10 -rP -p no:cacheprovider > $OUT/r3v36_tests.log 2>&1
echo "tests rc=$?" >> $LOG
grep -n "^_____.* test_\|^E  \|passed\|failed" $OUT/r3v36_tests.log | head -40 >> $LOG
grep "\[parity\]" $OUT/r3v36_tests.log | grep -i "c5 transformer\|odd" | cut -c1-200 >> $LOG
for MODE in 1 0 1 0; do
  echo "qkv_fuse=$MODE" >> $OUT/r3v36_diag.txt
  GM_DECODE_QKV_FUSE=$MODE timeout 300 python tools/diag_c5.py 2>&1 | grep -v amdgpu >> $OUT/r3v36_diag.txt
done
cat $OUT/r3v36_diag.txt >> $LOG
timeout 600 python tools/bench_c5.py > $OUT/r3v36_c5.json 2> $OUT/r3v36_c5.err; grep '^{' $OUT/r3v36_c5.json | head -c 400 >> $LOG
echo "done $(date)" >> $LOG
