"""GPU: does the 256 MiB Infinity Cache (MALL) serve the element-wise passes between the convolutions?  Times, per tensor size,
  (a) gn_apply x -> y repeated on the SAME x (reads can hit only if READS allocate), (b) the read-only statistics pass over a tensor the previous
  kernel has just WRITTEN (hits only if WRITES allocate), cold = the same pass over a tensor not touched for > 1 GB of other traffic,
  (c) gn_apply walking its rows in the order the producer wrote them vs in REVERSE (GM_GN_APPLY_REVERSE=1, a 268 MB tensor against a 256 MiB LRU).
usage: python tools/mall_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from generativemodels_amd import ops

dev = "cuda"


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


flush = torch.empty((1, 1 << 29), dtype=torch.bfloat16, device=dev)  # 1 GiB


def evict():
    flush.add_(1.0)


for edge, c in ((64, 64), (64, 128), (96, 64), (128, 64), (128, 128)):
    x = torch.randn((1, edge, edge, edge, c), device=dev).bfloat16()
    y = torch.empty_like(x)
    mb = x.numel() * 2 / 1e6
    sc, sh = torch.ones((1, c), device=dev), torch.zeros((1, c), device=dev)
    t_apply = timed(lambda: ops.gn_apply(x, sc, sh, "silu", out=y))
    # read-only pass right behind the writer of its input
    def write_then_read():
        ops.gn_apply(x, sc, sh, "silu", out=y)
        ops._fresh_channel_stats(y)
    t_pair = timed(write_then_read)
    t_read_warm = timed(lambda: ops._fresh_channel_stats(y))
    def cold_read():
        evict(); ops._fresh_channel_stats(y)
    t_cold = timed(cold_read) - timed(evict)
    print(f"{edge}^3 x {c} ({mb:6.1f} MB): gn_apply {t_apply * 1e3:7.1f} us = {2 * mb / t_apply / 1e3:5.2f} TB/s | stats pass: same tensor repeatedly {t_read_warm * 1e3:7.1f} us "
          f"= {mb / t_read_warm / 1e3:5.2f} TB/s, behind its writer {1e3 * (t_pair - t_apply):7.1f} us = {mb / max(t_pair - t_apply, 1e-6) / 1e3:5.2f} TB/s, "
          f"cold {t_cold * 1e3:7.1f} us = {mb / max(t_cold, 1e-6) / 1e3:5.2f} TB/s", flush=True)
