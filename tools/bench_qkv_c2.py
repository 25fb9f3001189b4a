"""GPU: the q | k | v projection of C2's mid-block attention (32 768 tokens, 256 -> 768 channels, GroupNorm prologue): the tile kernels (cfg 9 is what
the policy picks) vs the token GEMM (gm_linear_rows_affine) beyond its measured bounds.   usage: python tools/bench_qkv_c2.py"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from generativemodels_amd import ops
dev, dt = "cuda", torch.bfloat16
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows, cin, cout in [(32768, 256, 768), (32768, 256, 256), (4096, 256, 768)]:
    x = torch.randn((1, rows, cin), device=dev).to(dt)
    w = (torch.randn((cout, cin), device=dev) / math.sqrt(cin)).to(dt)
    b = torch.randn((cout,), device=dev)
    pre = (torch.rand((1, cin), device=dev) + 0.5, torch.randn((1, cin), device=dev) * 0.1)
    for name, kw in (("plain", {}), ("gn-prologue", dict(pre=pre, pre_act="none"))):
        keep = (ops.TOKEN_GEMM_MAX_ROWS, ops.TOKEN_GEMM_MAX_FLOP)
        ref = ops.linear(x, w, b, **kw)
        t_auto = timeit(lambda: ops.linear(x, w, b, **kw))
        ops.TOKEN_GEMM_MAX_ROWS, ops.TOKEN_GEMM_MAX_FLOP = 1 << 30, 1e30
        try:
            got = ops.linear(x, w, b, **kw)
            t_tok = timeit(lambda: ops.linear(x, w, b, **kw))
            err = (got.float() - ref.float()).abs().max().item()
        finally:
            ops.TOKEN_GEMM_MAX_ROWS, ops.TOKEN_GEMM_MAX_FLOP = keep
        line = f"{rows} x {cin}->{cout} {name:12s}: policy {t_auto:7.1f} us | token GEMM {t_tok:7.1f} us (max |diff| {err:.3e})"
        for cfg in (9, 10, 6, 1):
            try:
                line += f" | cfg{cfg} {timeit(lambda: ops.linear(x, w, b, force_cfg=cfg, **kw)):7.1f}"
            except Exception:
                line += f" | cfg{cfg} n/a"
        print(line, flush=True)
