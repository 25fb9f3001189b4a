"""GPU: where the cycles of one split-K slice launch (conv_sk.hip) go.  Thread 0 of every work-group stamps the shader clock at the phase
boundaries (debug_flags bit 12, slots behind the partial sums); prints the median over the work-groups per shape.
usage: python tools/sk_timeline.py"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from generativemodels_amd import ops

dev = "cuda"
NAMES = ["entry->decoded", "issue patch + 9 panels", "wait patch", "transform (prologue)", "wait panel 0 + barrier", "groups 0-3 (to barrier 4)",
         "groups 4-8", "shortcut", "partial stores issued", "own stores left"]


def run(cin, cout, size, ks, pre):
    x = torch.randn((1, size, size, size, cin), device=dev).bfloat16()
    w = (torch.randn((cout, cin, 3, 3, 3), device=dev) / math.sqrt(cin * 27)).bfloat16()
    b = torch.randn((cout,), device=dev)
    kw = dict(kernel=3, padding=1, force_cfg=11, want_stats=True, ksplit=ks)
    if pre:
        kw.update(pre=(torch.rand((1, cin), device=dev) + 0.5, torch.randn((1, cin), device=dev) * 0.1), pre_act="silu")
    keep = ops.DMA_FUSED_PROLOGUE
    ops.DMA_FUSED_PROLOGUE = "always"
    try:
        for _ in range(3):
            ops.conv(x, w, b, **kw)
        ops._CONV_DEBUG_FLAGS = 4096
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv(x, w, b, **kw); e1.record(); torch.cuda.synchronize()
    finally:
        ops._CONV_DEBUG_FLAGS = 0
        ops.DMA_FUSED_PROLOGUE = keep
    t = ops._SK_STAMPS.cpu().numpy()
    t = t[t[:, 0] > 0]
    d = np.diff(t[:, :11], axis=1)
    wall = (t[:, 15] - t[:, 14]) * 10.0  # ns (100 MHz)
    span_ns = (t[:, 15].max() - t[:, 14].min()) * 10.0
    cyc = t[:, 10] - t[:, 0]
    print(f"--- {cin}->{cout} at {size}^3, {ks} slices, prologue={pre}: {len(t)} work-groups; work-group life median {np.median(cyc):.0f} cycles = {np.median(wall):.0f} ns "
          f"({np.median(cyc) / max(np.median(wall), 1):.2f} GHz); first entry -> last exit {span_ns:.0f} ns; launch + combine by events {1e3 * e0.elapsed_time(e1):.1f} us")
    print(f"    entry spread (last - first work-group entry): {(t[:, 14].max() - t[:, 14].min()) * 10.0:.0f} ns")
    for i, nm in enumerate(NAMES):
        print(f"    {nm:28s} median {np.median(d[:, i]):8.0f}   p90 {np.percentile(d[:, i], 90):8.0f} cycles")


for cin, cout, size, ks, pre in ((256, 256, 8, 8, True), (128, 128, 16, 4, True), (64, 64, 32, 2, True), (64, 64, 32, 2, False), (192, 64, 32, 3, True), (512, 256, 8, 8, True)):
    run(cin, cout, size, ks, pre)
