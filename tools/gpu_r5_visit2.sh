#!/bin/bash
# round 5, GPU visit 2: sub-pixel kernel with both W parities per work item (cfg 17), conv_in on the fast epilogue (cfg 12), out head with two
# work-groups per CU (cfg 20, GM_CONV_COUT1_LTW=4)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=$OUT/r5v2.log; : > $L
step() { echo "== $1 ($(date +%T))" >> $L; }
step kernel-tests; timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider > $OUT/r5v2_ktests.log 2>&1; tail -6 $OUT/r5v2_ktests.log >> $L
step kernel-tests-ltw4; GM_CONV_COUT1_LTW=4 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "cout1 or cout or head or march" > $OUT/r5v2_ktests_ltw4.log 2>&1; tail -4 $OUT/r5v2_ktests_ltw4.log >> $L
step new-model-tests; timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_backward.py -q -p no:cacheprovider -k "selectable or captured or use_checkpointing or unet_bf16 or unet_fp32 or autoencoderkl" > $OUT/r5v2_mtests.log 2>&1; tail -6 $OUT/r5v2_mtests.log >> $L
bq() {
  TAGN=${1//[^A-Za-z0-9]/_}
  env $1 timeout 300 python bench.py --steps 3 --warmup 1 --cpu-baseline off 2> $OUT/r5v2_benchq_$TAGN.err | tail -1 > $OUT/r5v2_benchq_$TAGN.json
  python - $OUT/r5v2_benchq_$TAGN.json "$1" >> $L <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("benchq", sys.argv[2], "| vol/s", d["value"], "ms/iter", d["ms_per_ddim_iteration"], "fwd", d["unet_forward_ms"], "dominant", d["roofline"]["kernel"], d["roofline"]["achieved"], "avg ms", d["roofline"]["avg_launch_ms"], "J/vol", d["joules_per_volume"], "W", (d["package_power_w"] or {}).get("mean_w"))
    for k, v in list(d["kernel_breakdown_ms"].items())[:9]: print("   ", k, v)
except Exception as ex:
    print("benchq", sys.argv[2], "FAILED", ex)
PY
}
step benchq; bq "GM_CONV_COUT1_LTW=5"; bq "GM_CONV_COUT1_LTW=4"; bq "GM_CONV_COUT1_LTW=5"
step layers; timeout 300 python tools/layer_times.py > $OUT/r5v2_layer_times.txt 2>&1; tail -45 $OUT/r5v2_layer_times.txt >> $L
step ae256; timeout 300 python tools/layer_times_ae.py > $OUT/r5v2_layer_times_ae.txt 2>&1; tail -12 $OUT/r5v2_layer_times_ae.txt >> $L
step c3; timeout 300 python tools/bench_c3.py > $OUT/r5v2_c3.json 2> $OUT/r5v2_c3.err; grep '^{' $OUT/r5v2_c3.json | head -c 500 >> $L; echo >> $L
step full-tests; timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider --deselect tests/test_gpu_kernels.py > $OUT/r5v2_tests.log 2>&1; tail -8 $OUT/r5v2_tests.log >> $L
step done
tail -150 $L
