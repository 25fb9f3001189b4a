"""Fused bf16 attention backward (gm_attention_backward_fused) vs what the round-4 policy ran (composed bf16 path / fused fp32 kernels), GPU box only.
Arguments: BxHxLxdh or BxHxLqxLkxdh shapes.  Prints ms per call and TFLOP/s of the five-GEMM count (10 B H L^2 dh)."""
import math
import sys
import torch

sys.path.insert(0, ".")
from generativemodels_amd import autograd as A, ops  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


shapes = [(1, 1, 32768, 256), (1, 1, 16384, 256), (1, 1, 8192, 256), (1, 1, 4096, 256), (1, 1, 32768, 64), (1, 1, 8192, 128), (1, 1, 4096, 128)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for shp in shapes:
    b, h, l, dh = shp[0], shp[1], shp[2], shp[-1]
    lk = shp[3] if len(shp) == 5 else l  # BxHxLqxLkxdh: cross-attention shapes
    scale = 1 / math.sqrt(dh)
    g = torch.Generator().manual_seed(1)
    q, go = (torch.randn((b, l, h * dh), generator=g).bfloat16().cuda() for _ in range(2))
    k, v = (torch.randn((b, lk, h * dh), generator=g).bfloat16().cuda() for _ in range(2))
    lse = torch.empty((b, h, l), dtype=torch.float32, device="cuda") if ops.attention_writes_lse(q, k, v, h) else None
    o = ops.attention(q, k, v, h, scale, lse_out=lse)
    tf, outf = timed(lambda: ops.attention_backward_fused(q, k, v, o, go, h, scale))
    tg, outg = timed(lambda: ops.attention_backward_fused(q, k, v, o, go, h, scale, lse=lse)) if lse is not None else (float("nan"), outf)
    tc, outc = timed(lambda: A._attention_backward(q, k, v, o, go, h, scale))  # the round-4 policy
    tfw, _ = timed(lambda: ops.attention(q, k, v, h, scale))
    tfl, _ = timed(lambda: ops.attention(q, k, v, h, scale, lse_out=lse)) if lse is not None else (float("nan"), None)
    err = max((a.float() - c.float()).abs().max().item() / max(1e-6, c.float().abs().max().item()) for a, c in zip(outg, outc))
    fl = 10.0 * b * h * l * lk * dh
    print(f"B{b} H{h} L{l}x{lk} dh{dh}: fused {tf:8.3f} ms ({fl / tf / 1e9:6.1f} TFLOP/s of 5 GEMMs) | with the forward's LSE {tg:8.3f} ({fl / tg / 1e9:6.1f}) | round-4 policy {tc:8.3f} "
          f"({fl / tc / 1e9:6.1f}) | forward {tfw:7.3f} ms, writing LSE {tfl:7.3f} | rel diff vs round-4 path {err:.2e}", flush=True)
