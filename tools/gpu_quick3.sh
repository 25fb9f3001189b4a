#!/bin/bash
# one short visit: attention tests + variant timing, C3 under rocprofv3 kernel statistics, bench without the CPU leg
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -q -x -k "attention or transformer" -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/bench_attention.py 2>&1 | grep -v amdgpu | tail -12
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r2_prof_c3 -o c3 -- python $OLDPWD/tools/bench_c3.py > $OLDPWD/gpurun_out/r2_prof_c3.log 2>&1)
f=$(find gpurun_out/r2_prof_c3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -24 "$f" | cut -c1-230
find gpurun_out/r2_prof_c3 -name "*kernel_trace.csv" -size +20M -delete
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bench', d['value'], d['ms_per_ddim_iteration'], d['unet_forward_ms'], d['roofline']['achieved'], d['roofline']['frac'], d['kernel_breakdown_ms'].get('attention<bfloat16>'))"
