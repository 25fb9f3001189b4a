"""GPU micro-benchmark of the small 3x3x3 convolutions of a latent UNet (C3: 32^3 / 16^3 / 8^3 levels): tile configurations with and without
the fused GroupNorm prologue, plus the un-fused alternative gn_apply + prologue-free conv.   usage: python tools/bench_conv_small.py"""
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
for cin, cout, size in [(64, 64, 32), (128, 128, 16), (256, 128, 16), (256, 256, 8), (512, 256, 8)]:
    x = torch.randn((1, size, size, size, cin), device=dev).to(dt)
    w = (torch.randn((cout, cin, 3, 3, 3), device=dev) / math.sqrt(cin * 27)).to(dt)
    b = torch.randn((cout,), device=dev)
    pre = (torch.rand((1, cin), device=dev) + 0.5, torch.randn((1, cin), device=dev) * 0.1)
    line = f"{cin}->{cout}@{size}^3:"
    for cfg in (11, 14, 4, 2, 8, 9):
        try:
            line += f"  plain cfg{cfg} {timeit(lambda: ops.conv(x, w, b, kernel=3, padding=1, force_cfg=cfg)):6.1f}us"
        except Exception as e:
            line += f"  plain cfg{cfg}   n/a  "
    print(line)
    line = " " * len(f"{cin}->{cout}@{size}^3:")
    for cfg in (4, 2, 8, 9):
        try:
            line += f"  fused cfg{cfg} {timeit(lambda: ops.conv(x, w, b, kernel=3, padding=1, pre=pre, pre_act='silu', force_cfg=cfg)):6.1f}us"
        except Exception as e:
            line += f"  fused cfg{cfg}   n/a  "
    line += f"  | gn_apply {timeit(lambda: ops.gn_apply(x, pre[0], pre[1], 'silu')):5.1f}us   auto-fused {timeit(lambda: ops.conv(x, w, b, kernel=3, padding=1, pre=pre, pre_act='silu')):6.1f}us"
    print(line)
