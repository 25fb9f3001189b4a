#!/bin/bash
# round 3, GPU visit 27: token GEMM path of the 1x1 convolutions: kernel + model tests, C3 per-kernel durations, C4 step
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
LOG=$OUT/r3v27_round.log
echo "$(date)" > $LOG
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_backward.py -m gpu -q --maxfail=10 -p no:cacheprovider > $OUT/r3v27_tests.log 2>&1
echo "tests rc=$?" >> $LOG
grep -n "^_____.* test_\|^E  \|passed\|failed" $OUT/r3v27_tests.log | head -20 >> $LOG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/r3v27_prof -o c3 -- python $R/tools/layer_times_c3.py > $R/$OUT/r3v27_prof.log 2>&1)
F=$(find $OUT/r3v27_prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $OUT/r3v27_c3_kernel_stats.csv
rm -rf $OUT/r3v27_prof
timeout 300 python tools/bench_c3.py > $OUT/r3v27_c3.json 2> $OUT/r3v27_c3.err; grep '^{' $OUT/r3v27_c3.json | head -c 300 >> $LOG
timeout 600 python tools/bench_train.py > $OUT/r3v27_train.json 2> $OUT/r3v27_train.err; grep '^{' $OUT/r3v27_train.json | head -c 1500 >> $LOG
echo "done $(date)" >> $LOG
