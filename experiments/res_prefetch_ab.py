"""GPU: A/B of the residual prefetch of the LDS-DMA convolutions (gm_conv_dma_set_res_prefetch) on the C2 shapes that carry a residual, alternating
off / on four times per shape so that clock drift cancels.  usage: python tools/res_prefetch_ab.py"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from generativemodels_amd import ops
from generativemodels_amd._native import lib

dev = "cuda"
for name, cin, cout, edge in [("64->64@128^3 +res", 64, 64, 128), ("128->64@128^3 +res", 128, 64, 128), ("128->128@64^3 +res", 128, 128, 64), ("256->256@32^3 +res", 256, 256, 32),
                              ("64->64@256^3 +res", 64, 64, 256)]:
    x = torch.randn((1, edge, edge, edge, cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((cout, cin, 3, 3, 3), device=dev) / math.sqrt(cin * 27)).to(torch.bfloat16)
    b = torch.randn((cout,), device=dev)
    res = torch.randn((1, edge, edge, edge, cout), device=dev).to(torch.bfloat16)
    flops = 2.0 * edge ** 3 * cin * cout * 27
    kw = dict(kernel=3, padding=1, res=res, want_stats=True, force_cfg=14)
    n = 8 if edge == 256 else 30
    for _ in range(n):
        ops.conv(x, w, b, **kw)
    torch.cuda.synchronize()
    ref, times = None, {0: [], 1: []}
    for rep in range(4):
        for on in (0, 1):
            lib().gm_conv_dma_set_res_prefetch(on)
            y = ops.conv(x, w, b, **kw)
            if ref is None:
                ref = y.clone()
            assert torch.equal(y, ref)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                ops.conv(x, w, b, **kw)
            e1.record()
            torch.cuda.synchronize()
            times[on].append(e0.elapsed_time(e1) / n)
    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    nores = None
    kw2 = dict(kernel=3, padding=1, want_stats=True, force_cfg=14)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.conv(x, w, b, **kw2)
    e0.record()
    for _ in range(n):
        ops.conv(x, w, b, **kw2)
    e1.record()
    torch.cuda.synchronize()
    nores = e0.elapsed_time(e1) / n
    print(f"{name:22s} prefetch off {med[0]:.4f} ms ({flops / med[0] / 1e9:5.0f} TF/s)  on {med[1]:.4f} ms ({flops / med[1] / 1e9:5.0f} TF/s)  {100 * (med[0] / med[1] - 1):+.1f} %   "
          f"without a residual {nores:.4f} ms   runs off {[round(v, 4) for v in times[0]]} on {[round(v, 4) for v in times[1]]}", flush=True)
    del x, w, res, y, ref
lib().gm_conv_dma_set_res_prefetch(1)
