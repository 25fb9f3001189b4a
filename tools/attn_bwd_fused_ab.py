"""Fused bf16 attention backward (gm_attention_backward_fused) vs the composed bf16 path (score pass + three weight gradients), per kernel.
GPU box only.  Prints ms per call, TFLOP/s of the five-GEMM count (10 L^2 dh) and the per-kernel split from ops' timing hooks."""
import math
import sys
import torch

sys.path.insert(0, ".")
from generativemodels_amd import ops  # noqa: E402


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


shapes = [(32768, 256), (16384, 256), (8192, 256), (4096, 256), (32768, 64), (8192, 128), (4096, 128)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for l, dh in shapes:
    scale = 1 / math.sqrt(dh)
    g = torch.Generator().manual_seed(1)
    q, k, v, go = (torch.randn((1, l, dh), generator=g).bfloat16().cuda() for _ in range(4))
    o = ops.attention(q, k, v, 1, scale)
    lse = torch.zeros((1, 1, l), dtype=torch.float32, device="cuda")
    tf, outf = timed(lambda: ops.attention_backward_fused(q, k, v, o, go, 1, scale))
    tg, _ = timed(lambda: ops.attention_backward_fused(q, k, v, o, go, 1, scale, lse=lse))  # (wrong LSE: timing only)
    tc, outc = timed(lambda: ops.attention_backward_bf16(q, k, v, o, go, 1, scale))
    tfw, _ = timed(lambda: ops.attention(q, k, v, 1, scale))
    err = max((a.float() - b.float()).abs().max().item() / max(1e-6, b.float().abs().max().item()) for a, b in zip(outf, outc))
    fl = 10.0 * l * l * dh
    print(f"L {l} dh {dh}: fused {tf:8.3f} ms ({fl / tf / 1e9:6.1f} TFLOP/s of 5 GEMMs) | with caller LSE {tg:8.3f} ms ({fl / tg / 1e9:6.1f}) | composed {tc:8.3f} ms "
          f"({fl / tc / 1e9:6.1f}) | forward {tfw:7.3f} ms ({4.0 * l * l * dh / tfw / 1e9:6.1f}) | fused vs composed rel diff {err:.2e}", flush=True)
