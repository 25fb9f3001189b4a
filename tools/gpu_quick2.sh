#!/bin/bash
# one short visit: conv tests, tile-configuration timing table (cfg 11 vs 14 / 16 / 18), bench without the CPU leg, layer times
cd "$(dirname "$0")/.."
timeout 400 python -m pytest tests -m gpu -q -x -k "conv or upsample" -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/check_conv_cfgs.py 14,16,18 --time-only 2>&1 | grep -v amdgpu
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bench', d['value'], d['ms_per_ddim_iteration'], d['unet_forward_ms'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
timeout 200 python tools/layer_times.py 2>/dev/null | grep -E "cfg17|cfg15|sum of"
