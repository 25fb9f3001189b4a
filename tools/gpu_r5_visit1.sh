#!/bin/bash
# round 5, GPU visit 1: the phase-offset experiment + the register-direct epilogue A/B + clock / energy evidence
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=$OUT/r5v1.log; : > $L
step() { echo "== $1 ($(date +%T))" >> $L; }
step hwid; timeout 60 tools/hwid_probe.bin 4096 40000 0 >> $L 2>&1; timeout 60 tools/hwid_probe.bin 4096 40000 20 >> $L 2>&1
step kernel-tests; timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "conv" > $OUT/r5v1_ktests.log 2>&1; tail -5 $OUT/r5v1_ktests.log >> $L
step sweep-product; timeout 600 python tools/skew_sweep.py 14 > $OUT/r5v1_sweep_product.txt 2>&1; cat $OUT/r5v1_sweep_product.txt >> $L
step sweep-ldsep; GM_NATIVE_LIB=$PWD/generativemodels_amd/lib/libgmamd_ldsep.so timeout 600 python tools/skew_sweep.py 14 0,-1,16000,32000 > $OUT/r5v1_sweep_ldsep.txt 2>&1; cat $OUT/r5v1_sweep_ldsep.txt >> $L
step sweep-cfg11-17; SWEEP_SHAPES="64->64@128^3 +res,128->128@64^3" timeout 300 python tools/skew_sweep.py 11 0,-1,16000 >> $L 2>&1
step benchq
for ENVS in "GM_CONV_DMA_SKEW=0" "GM_CONV_DMA_SKEW=-1" "GM_CONV_DMA_SKEW=-1 GM_DMA_FUSED_PROLOGUE=always" "GM_CONV_DMA_SKEW=0 GM_DMA_FUSED_PROLOGUE=always" "GM_CONV_DMA_SKEW=-1 GM_CONV_DMA_GRID=-1"; do
  TAGN=${ENVS//[^A-Za-z0-9]/_}
  env $ENVS timeout 300 python bench.py --steps 3 --warmup 1 --cpu-baseline off 2> $OUT/r5v1_benchq_$TAGN.err | tail -1 > $OUT/r5v1_benchq_$TAGN.json
  python - $OUT/r5v1_benchq_$TAGN.json "$ENVS" >> $L <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("benchq", sys.argv[2], "| vol/s", d["value"], "ms/iter", d["ms_per_ddim_iteration"], "fwd", d["unet_forward_ms"], "dominant", d["roofline"]["kernel"], d["roofline"]["achieved"], "avg ms", d["roofline"]["avg_launch_ms"], "J/vol", d["joules_per_volume"], "W", (d["package_power_w"] or {}).get("mean_w"))
    for k, v in list(d["kernel_breakdown_ms"].items())[:8]: print("   ", k, v)
except Exception as ex:
    print("benchq", sys.argv[2], "FAILED", ex)
PY
done
step benchq-ldsep; GM_CONV_DMA_SKEW=0 GM_NATIVE_LIB=$PWD/generativemodels_amd/lib/libgmamd_ldsep.so timeout 300 python bench.py --steps 3 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ldsep skew 0: vol/s', d['value'], 'ms/iter', d['ms_per_ddim_iteration'], 'cfg14', d['roofline']['achieved'])" >> $L 2>&1
step energy; timeout 300 python tools/clock_energy.py energy > $OUT/r5v1_energy.jsonl 2>&1; cat $OUT/r5v1_energy.jsonl >> $L
step clock-pmc; rm -rf $OUT/r5v1_pmc; (cd /tmp && timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OLDPWD/$OUT/r5v1_pmc -o c -- python $OLDPWD/tools/clock_energy.py pmc > $OLDPWD/$OUT/r5v1_pmc.log 2>&1); python tools/clock_energy.py parse $OUT/r5v1_pmc > $OUT/r5v1_clock.json 2>> $L; head -c 3000 $OUT/r5v1_clock.json >> $L
find $OUT/r5v1_pmc -name "*.csv" -size +2M -delete 2>/dev/null
step full-tests; timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $OUT/r5v1_tests.log 2>&1; tail -8 $OUT/r5v1_tests.log >> $L
step done
tail -150 $L
