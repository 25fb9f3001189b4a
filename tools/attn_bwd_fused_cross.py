"""GPU: re-measurement of one line of profiles/r05_attn_bwd_fused_cross.txt: the fused bf16 attention backward WITH the forward's log-sum-exp at cross-attention
shapes -- B2 H8 L1024x256 dh64 once read 23.2 ms (3 repetitions, one warm-up) against 0.059 ms without the LSE.  Median and extremes of 50 repetitions per form.
usage: python tools/attn_bwd_fused_cross.py"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from generativemodels_amd import ops

dev = "cuda"
for b, h, lq, lk, dh in ((2, 8, 1024, 256, 64), (1, 8, 300, 300, 64), (1, 4, 256, 256, 128), (2, 8, 4096, 77, 64)):
    c = h * dh
    g = torch.Generator().manual_seed(3)
    q, go = (torch.randn((b, lq, c), generator=g).bfloat16().to(dev) for _ in range(2))
    k, v = (torch.randn((b, lk, c), generator=g).bfloat16().to(dev) for _ in range(2))
    scale = 1 / math.sqrt(dh)
    writes = ops.attention_writes_lse(q, k, v, h)
    lse = torch.empty((b, h, lq), dtype=torch.float32, device=dev) if writes else None
    o = ops.attention(q, k, v, h, scale, lse_out=lse)

    def reps(fn, n=50):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        return ts[len(ts) // 2], ts[0], ts[-1]

    own = reps(lambda: ops.attention_backward_fused(q, k, v, o, go, h, scale))
    line = f"B{b} H{h} L{lq}x{lk} dh{dh}: own LSE sweep median {own[0]:.3f} ms (min {own[1]:.3f}, max {own[2]:.3f})"
    if lse is not None:
        given = reps(lambda: ops.attention_backward_fused(q, k, v, o, go, h, scale, lse=lse))
        line += f" | with the forward's LSE median {given[0]:.3f} ms (min {given[1]:.3f}, max {given[2]:.3f})"
    else:
        line += " | the forward kernel writes no LSE at this shape"
    print(line, flush=True)
