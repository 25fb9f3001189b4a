"""A/B of ops.attention_backward_bf16 at long sequences: whole (sample, head) pair vs query slabs (ops.ATTENTION_BWD_BF16_SLAB_BYTES).
Prints ms per call and the allocator's peak scratch.  GPU box only."""
import math
import sys
import torch

sys.path.insert(0, ".")
from generativemodels_amd import ops  # noqa: E402


def run(l, dh, budget, reps=3):
    scale = 1 / math.sqrt(dh)
    g = torch.Generator().manual_seed(1)
    q, k, v, go = (torch.randn((1, l, dh), generator=g).bfloat16().cuda() for _ in range(4))
    o = ops.attention(q, k, v, 1, scale)
    old = ops.ATTENTION_BWD_BF16_SLAB_BYTES
    ops.ATTENTION_BWD_BF16_SLAB_BYTES = budget
    try:
        ops.attention_backward_bf16(q, k, v, o, go, 1, scale)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = ops.attention_backward_bf16(q, k, v, o, go, 1, scale)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, (torch.cuda.max_memory_allocated() - base) / 2**20, out
    finally:
        ops.ATTENTION_BWD_BF16_SLAB_BYTES = old


for l in (8192, 16384, 32768):
    for dh in (64,):
        a = run(l, dh, 64 << 30)
        for budget in (128 << 20, 512 << 20, 1 << 30):
            b = run(l, dh, budget)
            err = max((x.float() - y.float()).abs().max().item() for x, y in zip(a[2], b[2]))
            print(f"L {l} dh {dh}: whole {a[0]:8.2f} ms {a[1]:7.0f} MiB | slabs({budget >> 20} MiB) {b[0]:8.2f} ms {b[1]:7.0f} MiB | max|diff| {err:.3e}", flush=True)
