#!/bin/bash
# round 3, GPU visit 23: bf16-MFMA attention backward in the training path: backward tests, C4 gradients at real dims, C4 step
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
LOG=$OUT/r3v23_round.log
echo "$(date)" > $LOG
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize_oracle_r3.py tests/test_gpu_distributed.py -m gpu -q -k "not c3 and not c5 and not c1b" --maxfail=10 -rP -p no:cacheprovider > $OUT/r3v23_tests.log 2>&1
echo "tests rc=$?" >> $LOG
grep -n "^_____.* test_\|^E  \|passed\|failed" $OUT/r3v23_tests.log | head -30 >> $LOG
grep "\[parity\]" $OUT/r3v23_tests.log > $OUT/r3v23_parity.txt
timeout 400 python tools/bench_train.py 256 1 mixed 5 eager > $OUT/r3v23_train.json 2> $OUT/r3v23_train.err; echo "train rc=$?" >> $LOG; grep '^{' $OUT/r3v23_train.json | head -c 2500 >> $LOG
GM_ATTN_BWD_BF16_MIN_TOKENS=100000000 timeout 400 python tools/bench_train.py 256 1 mixed 5 eager > $OUT/r3v23_train_fp32attn.json 2> $OUT/r3v23_train_fp32attn.err; grep '^{' $OUT/r3v23_train_fp32attn.json | head -c 700 >> $LOG
timeout 300 python tools/try_graph_train.py > $OUT/r3v23_graph_c4.txt 2>&1; grep -v "amdgpu\|Warning\|Consider\|print(" $OUT/r3v23_graph_c4.txt >> $LOG
echo "done $(date)" >> $LOG
