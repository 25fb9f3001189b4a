#!/bin/bash
# round 3, GPU visit 2: whole suite (VQ-VAE training, determinism, marching out head, one-launch sub-pixel packs, reducer), A/B of the
# early-barrier conv variants, FETCH_SIZE calibration, bench.   logs in gpurun_out/r3v2_*
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
LOG=$OUT/r3v2_round.log
echo "$(date)" > $LOG
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 --durations=12 -rP -p no:cacheprovider > $OUT/r3v2_tests.log 2>&1
echo "tests rc=$?" >> $LOG
grep -n "^_____.* test_\|^E  \|passed\|failed" $OUT/r3v2_tests.log | head -60 >> $LOG
grep "\[parity\]" $OUT/r3v2_tests.log > $OUT/r3v2_parity.txt
L=$PWD/generativemodels_amd/lib
for V in main eb ebi main eb ebi; do
  if [ $V = main ]; then unset GM_NATIVE_LIB; else export GM_NATIVE_LIB=$L/libgmamd_$V.so; fi
  timeout 300 python tools/ab_lib.py $V >> $OUT/r3v2_ab.jsonl 2>> $OUT/r3v2_ab.err
done
unset GM_NATIVE_LIB
cat $OUT/r3v2_ab.jsonl >> $LOG
timeout 400 bash tools/fetch_calib.sh > $OUT/r3v2_fetch_calib.log 2>&1
echo "fetch_calib rc=$?" >> $LOG; tail -8 $OUT/r3v2_fetch_calib.log >> $LOG
timeout 600 python bench.py --cpu-baseline off > $OUT/r3v2_bench.json 2> $OUT/r3v2_bench.err
echo "bench rc=$?" >> $LOG; cat $OUT/r3v2_bench.json >> $LOG
timeout 300 python tools/layer_times.py > $OUT/r3v2_layer_times.txt 2>&1
GM_FORCE_REDUCER=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/bench_train.py 256 1 mixed 3 > $OUT/r3v2_train_mixed_rccl.json 2> $OUT/r3v2_train_mixed_rccl.err
echo "train mixed rccl rc=$?" >> $LOG; tail -c 1200 $OUT/r3v2_train_mixed_rccl.json >> $LOG; tail -3 $OUT/r3v2_train_mixed_rccl.err >> $LOG
timeout 300 python tools/bench_train.py 256 1 mixed 3 > $OUT/r3v2_train_mixed.json 2> $OUT/r3v2_train_mixed.err
echo "train mixed rc=$?" >> $LOG; tail -c 600 $OUT/r3v2_train_mixed.json >> $LOG
echo "done $(date)" >> $LOG
