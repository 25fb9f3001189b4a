#!/bin/bash
# One parameterised GPU visit (replaces the per-visit scripts of round 3): tools/gpu_visit.sh TAG STEP [STEP ...]
# Every step appends to gpurun_out/TAG_*.  Steps:
#   tests[:EXPR]    pytest -m gpu (optionally -k EXPR)            smoke       __graft_entry__.smoke()
#   bench           python bench.py (default run, CPU leg on)     benchq      bench.py --steps 3 --warmup 1 --cpu-baseline off
#   convab:CFGS     tools/bench_conv.py CFGS with BENCH_PLAIN=1   layers      tools/layer_times.py
#   ab:ENV=VAL      tools/ab_lib.py under that environment        c3 c5 c1b train   the other BASELINE configurations' benches
#   prof            rocprofv3 --kernel-trace --stats of the bench pmc  tools/pmc_traffic.sh
#   timeline:SHAPES tools/conv_timeline.py on the timeline build  sh:CMD      any shell command
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
TAG=$1; shift
LOG=$OUT/${TAG}_round.log
echo "== $TAG $(date)" >> $LOG
for STEP in "$@"; do
  NAME=${STEP%%:*}; ARG=""; [[ "$STEP" == *:* ]] && ARG=${STEP#*:}
  T0=$(date +%s)
  case $NAME in
    tests)
      if [ -n "$ARG" ]; then timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider -k "$ARG" > $OUT/${TAG}_tests.log 2>&1
      else timeout 1700 python -m pytest tests -m gpu -q --maxfail=30 --durations=8 -rP -p no:cacheprovider > $OUT/${TAG}_tests.log 2>&1; fi
      echo "tests[$ARG] rc=$?" >> $LOG
      grep -n "^_____.* test_\|^E  \|passed\|failed" $OUT/${TAG}_tests.log | head -40 >> $LOG
      grep "\[parity\]" $OUT/${TAG}_tests.log > $OUT/${TAG}_parity.txt ;;
    smoke) timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/${TAG}_smoke.log 2>&1; tail -2 $OUT/${TAG}_smoke.log >> $LOG ;;
    bench) timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?" >> $LOG; grep '^{' $OUT/${TAG}_bench.json >> $LOG ;;
    benchq) env $ARG timeout 300 python bench.py --steps 3 --warmup 1 --cpu-baseline off 2> $OUT/${TAG}_benchq.err | tail -1 > $OUT/${TAG}_benchq_${ARG//[^A-Za-z0-9]/_}.json
      python - "$OUT/${TAG}_benchq_${ARG//[^A-Za-z0-9]/_}.json" "$ARG" >> $LOG <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("benchq", sys.argv[2], d["value"], d["ms_per_ddim_iteration"], d["unet_forward_ms"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
      ;;
    convab) BENCH_PLAIN=1 timeout 300 python tools/bench_conv.py "$ARG" > $OUT/${TAG}_convab.txt 2>&1; cat $OUT/${TAG}_convab.txt >> $LOG ;;
    layers) env $ARG timeout 300 python tools/layer_times.py > $OUT/${TAG}_layer_times_${ARG//[^A-Za-z0-9]/_}.txt 2>&1; tail -1 $OUT/${TAG}_layer_times_${ARG//[^A-Za-z0-9]/_}.txt >> $LOG ;;
    ab) env $ARG timeout 300 python tools/ab_lib.py "$ARG" 2>/dev/null | tail -1 >> $OUT/${TAG}_ab.jsonl; tail -1 $OUT/${TAG}_ab.jsonl >> $LOG ;;
    c3) timeout 300 python tools/bench_c3.py > $OUT/${TAG}_c3.json 2> $OUT/${TAG}_c3.err; grep '^{' $OUT/${TAG}_c3.json | head -c 400 >> $LOG; echo >> $LOG ;;
    c5) timeout 300 python tools/bench_c5.py > $OUT/${TAG}_c5.json 2> $OUT/${TAG}_c5.err; grep '^{' $OUT/${TAG}_c5.json | head -c 400 >> $LOG; echo >> $LOG ;;
    c1b) timeout 300 python tools/bench_c1b.py > $OUT/${TAG}_c1b.json 2> $OUT/${TAG}_c1b.err; grep '^{' $OUT/${TAG}_c1b.json | head -c 600 >> $LOG; echo >> $LOG ;;
    train) timeout 600 python tools/bench_train.py > $OUT/${TAG}_train.json 2> $OUT/${TAG}_train.err; grep '^{' $OUT/${TAG}_train.json | head -c 900 >> $LOG; echo >> $LOG ;;
    prof)
      rm -rf $OUT/${TAG}_prof
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/${TAG}_prof -o p -- python $OLDPWD/bench.py --steps 1 --warmup 1 --cpu-baseline off > $OLDPWD/$OUT/${TAG}_prof.log 2>&1)
      F=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/${TAG}_kernel_stats.csv && head -8 $F >> $LOG ;;
    pmc) timeout 900 bash tools/pmc_traffic.sh >> $LOG 2>&1; cp $OUT/pmc_traffic/summary.json $OUT/${TAG}_pmc_traffic.json 2>/dev/null ;;
    timeline) GM_TL_SHAPES="$ARG" GM_NATIVE_LIB=$PWD/generativemodels_amd/lib/libgmamd_timeline.so timeout 300 python tools/conv_timeline.py > $OUT/${TAG}_timeline.txt 2>&1; cat $OUT/${TAG}_timeline.txt >> $LOG ;;
    sh) bash -c "$ARG" >> $LOG 2>&1 ;;
    *) echo "unknown step $STEP" >> $LOG ;;
  esac
  echo "-- $STEP: $(( $(date +%s) - T0 )) s" >> $LOG
done
echo "done $(date)" >> $LOG
tail -60 $LOG
