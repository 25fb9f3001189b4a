#!/bin/bash
# round 5, GPU visit 4: tile configuration 23 (conv_w4.hip: four waves of 4 x 2 blocks of the 32x32x16 MFMA) -- parity, then A/B against cfg 14 / 22
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=$OUT/r5v4.log; : > $L
step() { echo "== $1 ($(date +%T))" >> $L; }
step kernel-tests; timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "512_voxel" > $OUT/r5v4_ktests.log 2>&1; tail -12 $OUT/r5v4_ktests.log >> $L
step model-tests; timeout 600 python -m pytest tests/test_gpu_models.py -q -p no:cacheprovider -k "selectable" > $OUT/r5v4_mtests.log 2>&1; tail -6 $OUT/r5v4_mtests.log >> $L
step convab; BENCH_PLAIN=1 timeout 300 python tools/bench_conv.py 14,22,23 > $OUT/r5v4_convab.txt 2>&1; cat $OUT/r5v4_convab.txt >> $L
step convab-again; BENCH_PLAIN=1 timeout 300 python tools/bench_conv.py 23,14,23,14 > $OUT/r5v4_convab2.txt 2>&1; cat $OUT/r5v4_convab2.txt >> $L
bq() {
  TAGN=${1//[^A-Za-z0-9]/_}
  env $1 timeout 300 python bench.py --steps 3 --warmup 1 --cpu-baseline off 2> $OUT/r5v4_benchq_$TAGN.err | tail -1 > $OUT/r5v4_benchq_$TAGN.json
  python - $OUT/r5v4_benchq_$TAGN.json "$1" >> $L <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("benchq", sys.argv[2], "| vol/s", d["value"], "ms/iter", d["ms_per_ddim_iteration"], "fwd", d["unet_forward_ms"], "dominant", d["roofline"]["kernel"], d["roofline"]["achieved"], "avg ms", d["roofline"]["avg_launch_ms"], "J/vol", d["joules_per_volume"], "W", (d["package_power_w"] or {}).get("mean_w"))
    for k, v in list(d["kernel_breakdown_ms"].items())[:9]: print("   ", k, v)
except Exception as ex:
    print("benchq", sys.argv[2], "FAILED", ex, open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
}
step benchq; bq "GM_CONV_W4=0"; bq "GM_CONV_W4=1"; bq "GM_CONV_W4=0"; bq "GM_CONV_W4=1"
step layers-w4; GM_CONV_W4=1 timeout 300 python tools/layer_times.py > $OUT/r5v4_layer_times_w4.txt 2>&1; tail -46 $OUT/r5v4_layer_times_w4.txt | grep "conv_igemm\|sum of" >> $L
step ae256-w4; GM_CONV_W4=1 timeout 300 python tools/layer_times_ae.py > $OUT/r5v4_layer_times_ae_w4.txt 2>&1; tail -12 $OUT/r5v4_layer_times_ae_w4.txt >> $L
step done
tail -120 $L
