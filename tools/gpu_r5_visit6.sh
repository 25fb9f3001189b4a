#!/bin/bash
# round 5, GPU visit 6: producer / consumer tile order A/B (gn_apply XCD-owned eighths x convolution walk direction)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=$OUT/r5v6.log; : > $L
bq() {
  env $1 timeout 300 python bench.py --steps 3 --warmup 1 --cpu-baseline off 2> $OUT/r5v6_benchq.err | tail -1 > $OUT/r5v6_benchq.json
  python - $OUT/r5v6_benchq.json "$1" >> $L <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    kb = d["kernel_breakdown_ms"]
    print("benchq", sys.argv[2], "| vol/s", d["value"], "ms/iter", d["ms_per_ddim_iteration"], "fwd", d["unet_forward_ms"], "cfg14", kb["conv_igemm<bfloat16,cfg14>"]["ms"], "gn_apply", kb["gn_apply<bfloat16>"]["ms"], "cfg17", kb["conv_igemm<bfloat16,cfg17>"]["ms"], "attn", kb["attention<bfloat16>"]["ms"], "W", (d["package_power_w"] or {}).get("mean_w"))
except Exception as ex:
    print("benchq", sys.argv[2], "FAILED", ex)
PY
}
for rep in 1 2; do
  bq "GM_GN_APPLY_ORDER=0 GM_CONV_DMA_WALK_BACK=0"
  bq "GM_GN_APPLY_ORDER=2 GM_CONV_DMA_WALK_BACK=1"
  bq "GM_GN_APPLY_ORDER=0 GM_CONV_DMA_WALK_BACK=1"
  bq "GM_GN_APPLY_ORDER=1 GM_CONV_DMA_WALK_BACK=0"
done
echo "== tests under the new order" >> $L
GM_GN_APPLY_ORDER=2 GM_CONV_DMA_WALK_BACK=1 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -x -p no:cacheprovider -k "conv or gn or group or fullsize or chain" > $OUT/r5v6_tests.log 2>&1; tail -3 $OUT/r5v6_tests.log >> $L
cat $L
