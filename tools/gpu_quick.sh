#!/bin/bash
# one short visit: conv tests (kernel + full-size), bench without the CPU leg, cycle timeline of the LDS-DMA kernel
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests -m gpu -q -x -k "conv or upsample or fullsize or full_size" -p no:cacheprovider 2>&1 | tail -3
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bench', d['value'], d['ms_per_ddim_iteration'], d['unet_forward_ms'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
GM_TL_SHAPES="${GM_TL_SHAPES:-64,64,128,14;192,64,128,14}" GM_NATIVE_LIB=$PWD/generativemodels_amd/lib/libgmamd_timeline.so timeout 200 python tools/conv_timeline.py 2>/dev/null
