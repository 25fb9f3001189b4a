"""GPU: the LDS-DMA flash-attention kernel variants (queries per wave x key slices) at the shapes the BASELINE configurations use:
C2 mid block (1 head x 256, 32768 tokens), C3 latent UNet levels (16^3 tokens x 128, 8^3 x 256), plus a multi-head case.
usage: python tools/bench_attention.py"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from generativemodels_amd import ops, _native

dev = "cuda"
SHAPES = [("C2 mid 1x256 L=32768", 1, 1, 32768, 256), ("C3 lvl1 1x128 L=4096", 1, 1, 4096, 128), ("C3 lvl2 1x256 L=512", 1, 1, 512, 256),
          ("8 heads x 64, L=8192", 1, 8, 8192, 64), ("B2 1x256 L=16384", 2, 1, 16384, 256)]
VARIANTS = [(0, 0), (1, 1), (2, 1), (1, 2), (2, 2), (2, 4), (1, 4), (1, 8), (2, 8)]
for name, b, h, L, dh in SHAPES:
    c = h * dh
    g = torch.Generator(device=dev).manual_seed(1)
    qkv = torch.randn((b, L, 3 * c), generator=g, device=dev).bfloat16()
    res = torch.randn((b, L, c), generator=g, device=dev).bfloat16()
    scale = 1 / math.sqrt(dh)
    flops = 4.0 * b * h * L * L * dh
    line = f"{name:26s}"
    ref = None
    for qf, sp in VARIANTS:
        _native.lib().gm_attention_dma_set_variant(qf, sp)
        try:
            out = ops.attention(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], h, scale, res=res)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.float()
            err = (out.float() - ref).abs().max().item()
            n = 5 if L >= 16384 else 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                ops.attention(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], h, scale, res=res)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            line += f" | qf{qf}s{sp}: {ms:7.3f} ms {flops / ms / 1e9:5.0f} TF/s (d {err:.1e})"
        except Exception as ex:
            line += f" | qf{qf}s{sp}: n/a {str(ex)[:30]}"
        finally:
            _native.lib().gm_attention_dma_set_variant(0, 0)
    print(line, flush=True)
