#!/bin/bash
# round 3, GPU visit 1: the whole -m gpu suite (new real-size parity + mixed-precision + ControlNet-training tests), C4 training step in the
# three precisions, a baseline bench line.   logs in gpurun_out/r3v1_*.log
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
LOG=$OUT/r3v1_round.log
echo "$(date)" > $LOG
rocminfo 2>/dev/null | grep -m2 -E "gfx|Compute Unit" >> $LOG
nproc >> $LOG; free -g | head -2 >> $LOG
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 --durations=20 -rP -p no:cacheprovider > $OUT/r3v1_tests.log 2>&1
echo "tests rc=$?" >> $LOG
grep -v "^\[parity\]\|^---\|^$\|Captured\|^_____\|PASSED" $OUT/r3v1_tests.log | tail -80 >> $LOG
grep "\[parity\]" $OUT/r3v1_tests.log > $OUT/r3v1_parity.txt
for M in mixed bf16; do
  timeout 300 python tools/bench_train.py 256 1 $M 3 > $OUT/r3v1_train_$M.json 2> $OUT/r3v1_train_$M.err
  echo "train $M rc=$?" >> $LOG; tail -c 1500 $OUT/r3v1_train_$M.json >> $LOG; tail -3 $OUT/r3v1_train_$M.err >> $LOG
done
GM_FORCE_REDUCER=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/bench_train.py 256 1 mixed 3 > $OUT/r3v1_train_mixed_rccl.json 2> $OUT/r3v1_train_mixed_rccl.err
echo "train mixed rccl rc=$?" >> $LOG; tail -c 1500 $OUT/r3v1_train_mixed_rccl.json >> $LOG; tail -3 $OUT/r3v1_train_mixed_rccl.err >> $LOG
timeout 600 python bench.py --cpu-baseline off > $OUT/r3v1_bench.json 2> $OUT/r3v1_bench.err
echo "bench rc=$?" >> $LOG; cat $OUT/r3v1_bench.json >> $LOG
timeout 300 python tools/layer_times.py > $OUT/r3v1_layer_times.txt 2>&1
echo "layer_times rc=$?" >> $LOG
echo "done $(date)" >> $LOG
