#!/bin/bash
# A/B of the LDS-DMA grid policy on one box: bench (no CPU leg) + cycle timelines; optional first argument "t" runs the conv tests first
cd "$(dirname "$0")/.."
if [[ "${1:-}" == *t* ]]; then
  timeout 400 python -m pytest tests -m gpu -q -x -k conv -p no:cacheprovider 2>&1 | tail -3
fi
for G in 0 -1; do
  echo "== grid policy $G"
  GM_CONV_DMA_GRID=$G timeout 300 python bench.py --steps 3 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_ddim_iteration'], d['unet_forward_ms'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done
for G in 0 -1; do
echo "== timeline, policy $G"
GM_CONV_DMA_GRID=$G GM_TL_SHAPES="${GM_TL_SHAPES:-64,64,128,11;192,64,128,11}" GM_NATIVE_LIB=$PWD/generativemodels_amd/lib/libgmamd_timeline.so timeout 200 python tools/conv_timeline.py 2>/dev/null
done
