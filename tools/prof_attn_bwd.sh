#!/bin/bash
# rocprofv3 --kernel-trace --stats of the fused attention backward at C2's size (5 calls with the forward's LSE, 5 with the own LSE sweep)
cd "$(dirname "$0")/.."
R=$PWD; OUT=$R/gpurun_out/prof_attn_bwd; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/attn_bwd_reps.py <<'PY'
import math, sys, torch
sys.path.insert(0, sys.argv[1])
from generativemodels_amd import ops
l, dh = 32768, 256
g = torch.Generator().manual_seed(1)
q, k, v, go = (torch.randn((1, l, dh), generator=g).bfloat16().cuda() for _ in range(4))
lse = torch.empty((1, 1, l), dtype=torch.float32, device="cuda")
s = 1 / math.sqrt(dh)
for _ in range(5):
    o = ops.attention(q, k, v, 1, s, lse_out=lse)
    ops.attention_backward_fused(q, k, v, o, go, 1, s, lse=lse)
    ops.attention_backward_fused(q, k, v, o, go, 1, s)
torch.cuda.synchronize()
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python /tmp/attn_bwd_reps.py $R > $OUT/run.log 2>&1)
F=$(find $OUT -name "*kernel_stats.csv" | head -1); head -12 $F | cut -c1-210
find $OUT -name "*.csv" -size +1M -delete 2>/dev/null; find $OUT -name "*.db" -delete 2>/dev/null
