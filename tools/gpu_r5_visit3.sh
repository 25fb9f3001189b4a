#!/bin/bash
# round 5, GPU visit 3: MFMA energy per FLOP by instruction shape, residual prefetch A/B, the fixed tests, rocprofv3 stats of the bench
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=$OUT/r5v3.log; : > $L
step() { echo "== $1 ($(date +%T))" >> $L; }
step mfma-energy; timeout 200 tools/mfma_energy.bin > $OUT/r5v3_mfma_energy.txt 2>&1; cat $OUT/r5v3_mfma_energy.txt >> $L
step prefetch-ab; timeout 300 python tools/res_prefetch_ab.py > $OUT/r5v3_res_prefetch_ab.txt 2>&1; cat $OUT/r5v3_res_prefetch_ab.txt >> $L
step tests; timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_backward.py tests/test_gpu_kernels.py -q -p no:cacheprovider -k "selectable or use_checkpointing or conv_lds_dma or cout1 or march or head or fused_shortcut or residual" > $OUT/r5v3_tests.log 2>&1; tail -5 $OUT/r5v3_tests.log >> $L
bq() {
  TAGN=${1//[^A-Za-z0-9]/_}
  env $1 timeout 300 python bench.py --steps 3 --warmup 1 --cpu-baseline off 2> $OUT/r5v3_benchq_$TAGN.err | tail -1 > $OUT/r5v3_benchq_$TAGN.json
  python - $OUT/r5v3_benchq_$TAGN.json "$1" >> $L <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("benchq", sys.argv[2], "| vol/s", d["value"], "ms/iter", d["ms_per_ddim_iteration"], "fwd", d["unet_forward_ms"], "dominant", d["roofline"]["kernel"], d["roofline"]["achieved"], "avg ms", d["roofline"]["avg_launch_ms"], "J/vol", d["joules_per_volume"], "W", (d["package_power_w"] or {}).get("mean_w"))
    for k, v in list(d["kernel_breakdown_ms"].items())[:9]: print("   ", k, v)
except Exception as ex:
    print("benchq", sys.argv[2], "FAILED", ex)
PY
}
step benchq; bq "GM_CONV_DMA_RES_PREFETCH=0"; bq "GM_CONV_DMA_RES_PREFETCH=1"; bq "GM_CONV_DMA_RES_PREFETCH=0"; bq "GM_CONV_DMA_RES_PREFETCH=1"
step done
tail -120 $L
