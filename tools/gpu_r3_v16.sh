#!/bin/bash
# round 3, GPU visit 16: HIP-graph replay of the training step: tests (single GPU, forced RCCL reducer), C4 bench eager vs graph
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
LOG=$OUT/r3v16_round.log
echo "$(date)" > $LOG
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_distributed.py -m gpu -q -k "graphed or rccl or RCCL or reducer" --maxfail=5 -rP -p no:cacheprovider > $OUT/r3v16_tests.log 2>&1
echo "tests rc=$?" >> $LOG
grep -n "^_____.* test_\|^E  \|passed\|failed\|RCCL_WORKER" $OUT/r3v16_tests.log | head -40 >> $LOG
timeout 400 python tools/bench_train.py 256 1 mixed 5 graph > $OUT/r3v16_train_graph.json 2> $OUT/r3v16_train_graph.err; echo "graph rc=$?" >> $LOG; tail -c 1500 $OUT/r3v16_train_graph.json >> $LOG; tail -5 $OUT/r3v16_train_graph.err >> $LOG
timeout 400 python tools/bench_train.py 256 1 mixed 5 eager > $OUT/r3v16_train_eager.json 2> $OUT/r3v16_train_eager.err; echo "eager rc=$?" >> $LOG; tail -c 700 $OUT/r3v16_train_eager.json | head -c 700 >> $LOG
GM_FORCE_REDUCER=1 timeout 400 python tools/bench_train.py 256 1 mixed 5 graph > $OUT/r3v16_train_graph_rccl.json 2> $OUT/r3v16_train_graph_rccl.err; echo "graph rccl rc=$?" >> $LOG; grep '^{' $OUT/r3v16_train_graph_rccl.json | head -c 700 >> $LOG; tail -3 $OUT/r3v16_train_graph_rccl.err >> $LOG
echo "done $(date)" >> $LOG
