#!/bin/bash
# round 5, GPU visit 5: gn_apply block order A/B (XCD-owned eighths walked backwards), the other configurations' benches, the whole GPU suite
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=$OUT/r5v5.log; : > $L
step() { echo "== $1 ($(date +%T))" >> $L; }
bq() {
  TAGN=${1//[^A-Za-z0-9]/_}
  env $1 timeout 300 python bench.py --steps 3 --warmup 1 --cpu-baseline off 2> $OUT/r5v5_benchq_$TAGN.err | tail -1 > $OUT/r5v5_benchq_$TAGN.json
  python - $OUT/r5v5_benchq_$TAGN.json "$1" >> $L <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("benchq", sys.argv[2], "| vol/s", d["value"], "ms/iter", d["ms_per_ddim_iteration"], "fwd", d["unet_forward_ms"], "dominant", d["roofline"]["kernel"], d["roofline"]["achieved"], "avg ms", d["roofline"]["avg_launch_ms"], "J/vol", d["joules_per_volume"], "W", (d["package_power_w"] or {}).get("mean_w"))
    for k, v in list(d["kernel_breakdown_ms"].items())[:5]: print("   ", k, v)
except Exception as ex:
    print("benchq", sys.argv[2], "FAILED", ex)
PY
}
step benchq; bq "GM_GN_APPLY_ORDER=0"; bq "GM_GN_APPLY_ORDER=1"; bq "GM_GN_APPLY_ORDER=0"; bq "GM_GN_APPLY_ORDER=1"
step c3; timeout 300 python tools/bench_c3.py > $OUT/r5v5_c3.json 2> $OUT/r5v5_c3.err; grep '^{' $OUT/r5v5_c3.json | head -c 600 >> $L; echo >> $L
step c3-order1; GM_GN_APPLY_ORDER=1 timeout 300 python tools/bench_c3.py 2>/dev/null | grep '^{' | head -c 300 >> $L; echo >> $L
step c5; timeout 300 python tools/bench_c5.py > $OUT/r5v5_c5.json 2> $OUT/r5v5_c5.err; grep '^{' $OUT/r5v5_c5.json | head -c 400 >> $L; echo >> $L
step train; timeout 600 python tools/bench_train.py > $OUT/r5v5_train.json 2> $OUT/r5v5_train.err; grep '^{' $OUT/r5v5_train.json | head -c 900 >> $L; echo >> $L
step c1b; timeout 300 python tools/bench_c1b.py > $OUT/r5v5_c1b.json 2> $OUT/r5v5_c1b.err; grep '^{' $OUT/r5v5_c1b.json | head -c 500 >> $L; echo >> $L
step ae256; timeout 300 python tools/layer_times_ae.py > $OUT/r5v5_layer_times_ae.txt 2>&1; tail -24 $OUT/r5v5_layer_times_ae.txt >> $L
step full-tests; timeout 1700 python -m pytest tests -m gpu -q --maxfail=20 --durations=6 -p no:cacheprovider > $OUT/r5v5_tests.log 2>&1; tail -14 $OUT/r5v5_tests.log >> $L
step done
tail -120 $L
