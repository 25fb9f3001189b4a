"""GPU: the one-time phase offset between the co-resident work-groups of a CU (gm_conv_dma_set_phase_skew, conv_dma.hip) swept over the C2
convolution shapes on tile configuration 14, under both grid policies.  Prints ms / TFLOP/s per (shape, policy, skew); -1 = the automatic
choice.  usage: python tools/skew_sweep.py [cfg=14] [skews=0,-1,8000,...]   (GM_NATIVE_LIB selects a build variant)"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from generativemodels_amd import ops
from generativemodels_amd._native import lib

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 14
skews = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,-1,8000,16000,24000,32000,48000,64000".split(","))]
SHAPES = [("64->64@128^3 +res", 64, 64, 128, True), ("64->64@128^3", 64, 64, 128, False), ("128->64@128^3", 128, 64, 128, False), ("192->64@128^3", 192, 64, 128, False),
          ("128->128@64^3", 128, 128, 64, True), ("256->128@64^3", 256, 128, 64, False), ("384->128@64^3", 384, 128, 64, False), ("256->256@32^3", 256, 256, 32, True)]
if os.environ.get("SWEEP_SHAPES"):
    keep = os.environ["SWEEP_SHAPES"].split(",")
    SHAPES = [s for s in SHAPES if any(k in s[0] for k in keep)]
dev = "cuda"
print(f"# lib {os.environ.get('GM_NATIVE_LIB', 'product')}, cfg {cfg}")
for name, cin, cout, edge, with_res in SHAPES:
    x = torch.randn((1, edge, edge, edge, cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((cout, cin, 3, 3, 3), device=dev) / math.sqrt(cin * 27)).to(torch.bfloat16)
    b = torch.randn((cout,), device=dev)
    res = torch.randn((1, edge, edge, edge, cout), device=dev).to(torch.bfloat16) if with_res else None
    flops = 2.0 * edge ** 3 * cin * cout * 27
    ref = None
    for policy in (0, -1):
        lib().gm_conv_dma_set_persistent(policy)
        line = f"{name:20s} grid {policy:2d}"
        for sk in skews:
            lib().gm_conv_dma_set_phase_skew(sk)
            kw = dict(kernel=3, padding=1, res=res, want_stats=True, force_cfg=cfg)
            y = ops.conv(x, w, b, **kw)
            torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
            else:
                assert torch.equal(y, ref), (name, policy, sk)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _rep in range(2):
                e0.record()
                for _ in range(20):
                    ops.conv(x, w, b, **kw)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20)
            line += f" | {sk:6d}: {best:6.3f} ms {flops / best / 1e9:6.0f}"
        print(line, flush=True)
    del x, w, y, res
lib().gm_conv_dma_set_persistent(0)
lib().gm_conv_dma_set_phase_skew(-1)
