"""GPU: ablation of the fast conv kernel (debug_flags) on the 64->64@128^3 shape -- which phase costs what?"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from generativemodels_amd import ops
dev = "cuda"
def run(cin, cout, sp, pro, flags, cfg):
    x = torch.randn((1, *sp, cin), device=dev).bfloat16()
    w = (torch.randn((cout, cin, 3, 3, 3), device=dev) / math.sqrt(cin * 27)).bfloat16()
    b = torch.randn((cout,), device=dev)
    pre = (torch.rand((1, cin), device=dev) + 0.5, torch.randn((1, cin), device=dev) * 0.1) if pro else None
    ops._CONV_DEBUG_FLAGS = flags
    kw = dict(kernel=3, padding=1, pre=pre, pre_act="silu" if pro else "none", force_cfg=cfg)
    ops.conv(x, w, b, **kw); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ops.conv(x, w, b, **kw)
    e1.record(); torch.cuda.synchronize()
    ops._CONV_DEBUG_FLAGS = 0
    return e0.elapsed_time(e1) / 3
names = {0: "full", 1: "no A restage", 2: "no B loads", 4: "no MFMA (ds_reads kept)", 8: "no ds_read+MFMA", 16: "no epilogue",
         1 | 2: "no A restage, no B loads", 1 | 2 | 8: "only barriers+epilogue", 1 | 2 | 8 | 16: "skeleton", 4 | 1: "no MFMA no A restage", 8 | 2: "A staging only", 32: "old scatter epilogue"}
for cin, cout, sp, cfg in [(64, 64, (128, 128, 128), 7), (128, 128, (64, 64, 64), 6), (192, 64, (128, 128, 128), 7), (64, 64, (128, 128, 128), 5)]:
    for pro in (True, False):
        print(f"--- {cin}->{cout}@{sp} cfg{cfg} prologue={pro}")
        for f, nm in names.items():
            print(f"  flags {f:2d} {nm:28s}: {run(cin, cout, sp, pro, f, cfg):7.3f} ms", flush=True)
