"""GPU micro-benchmark of the 1x1 convolutions / token GEMMs of the latent-resolution attention blocks (C3 / C4: 4096 or 512 tokens, 128 or 256
channels): every tile configuration that accepts them, with and without the GroupNorm prologue.   usage: python tools/bench_conv1x1.py"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from generativemodels_amd import ops
dev, dt = "cuda", torch.bfloat16
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows, cin, cout in [(4096, 128, 384), (4096, 128, 128), (512, 256, 768), (512, 256, 256), (32768, 64, 64)]:
    x = torch.randn((1, rows, cin), device=dev).to(dt)
    w = (torch.randn((cout, cin), device=dev) / math.sqrt(cin)).to(dt)
    b = torch.randn((cout,), device=dev)
    pre = (torch.rand((1, cin), device=dev) + 0.5, torch.randn((1, cin), device=dev) * 0.1)
    for name, kw in (("plain", {}), ("gn-prologue", dict(pre=pre, pre_act="none"))):
        line = f"{rows} x {cin}->{cout} {name:12s}: auto {timeit(lambda: ops.linear(x, w, b, **kw)):6.1f}us"
        for cfg in range(0, 11):
            try:
                line += f"  cfg{cfg} {timeit(lambda: ops.linear(x, w, b, force_cfg=cfg, **kw)):6.1f}"
            except Exception as e:
                line += f"  cfg{cfg}   n/a "
        print(line, flush=True)
