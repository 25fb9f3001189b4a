#!/bin/bash
# One gpurun visit: GPU parity tests, smoke, bench, rocprof -- every stage under its own timeout, logs in gpurun_out/.
# usage: tools/gpu_round.sh [stages]   stages: any of k(ernels) m(odels) s(moke) b(ench) p(rofile) c(ounters)
set -u
cd "$(dirname "$0")/.."
STAGES="${1:-kmsbp}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
echo "stages=$STAGES $(date)" > $OUT/round.log
rocminfo 2>/dev/null | grep -m2 -E "gfx|Compute Unit" >> $OUT/round.log
nproc >> $OUT/round.log
if [[ $STAGES == *k* ]]; then
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -n 1 --maxfail=400 -p no:cacheprovider > $OUT/kernels.log 2>&1
  echo "kernels rc=$?" >> $OUT/round.log; tail -3 $OUT/kernels.log >> $OUT/round.log
fi
if [[ $STAGES == *m* ]]; then
  timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -n 1 --maxfail=400 -p no:cacheprovider > $OUT/models.log 2>&1
  echo "models rc=$?" >> $OUT/round.log; tail -3 $OUT/models.log >> $OUT/round.log
fi
if [[ $STAGES == *s* ]]; then
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
  echo "smoke rc=$?" >> $OUT/round.log; tail -2 $OUT/smoke.log >> $OUT/round.log
fi
if [[ $STAGES == *b* ]]; then
  timeout 300 python bench.py --size 32 --steps 1 --warmup 1 --inference-steps 4 --cpu-baseline small --graph 0 > $OUT/bench_small.log 2>&1
  echo "bench_small rc=$?" >> $OUT/round.log; tail -c 1500 $OUT/bench_small.log >> $OUT/round.log
  timeout 1200 python bench.py --steps 2 --warmup 1 > $OUT/bench.log 2>&1
  echo "bench rc=$?" >> $OUT/round.log; tail -c 3000 $OUT/bench.log >> $OUT/round.log
fi
if [[ $STAGES == *l* ]]; then
  timeout 600 python tools/layer_times.py > $OUT/layer_times.log 2>&1
  echo "layer_times rc=$?" >> $OUT/round.log; tail -1 $OUT/layer_times.log >> $OUT/round.log
fi
if [[ $STAGES == *p* ]]; then
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o bench -- python $OLDPWD/bench.py --steps 1 --warmup 0 --graph 0 --cpu-baseline off > $OLDPWD/$OUT/prof.log 2>&1)
  echo "prof rc=$?" >> $OUT/round.log
  find $OUT/prof -name "*kernel_stats*" | head -3 >> $OUT/round.log
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" >> $OUT/round.log
  # keep the trace small: stats only
  find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
fi
if [[ $STAGES == *c* ]]; then
  (cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE -d $OLDPWD/$OUT/pmc_fetch -o bench -- python $OLDPWD/bench.py --steps 1 --warmup 0 --graph 0 --cpu-baseline off --inference-steps 2 > $OLDPWD/$OUT/pmc_fetch.log 2>&1)
  echo "pmc_fetch rc=$?" >> $OUT/round.log
  (cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE -d $OLDPWD/$OUT/pmc_write -o bench -- python $OLDPWD/bench.py --steps 1 --warmup 0 --graph 0 --cpu-baseline off --inference-steps 2 > $OLDPWD/$OUT/pmc_write.log 2>&1)
  echo "pmc_write rc=$?" >> $OUT/round.log
fi
cat $OUT/round.log
