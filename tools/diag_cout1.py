"""GPU diagnostics of the C_out = 1 kernels: (1) where the marching kernel (cfg 20) and the tile kernel (cfg 13) differ in fp32 on the shapes
of tests/test_gpu_kernels.py; (2) time of both on the C2 out head (64 -> 1 at 128^3, bf16) with and without the fused GN + SiLU prologue and
for every depth-segment length (GM_CONV_COUT1_LTD is read at import: one process per value, see tools/gpu_r3_v3.sh)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from generativemodels_amd import ops

dev = "cuda"
def rnd(shape, seed, dtype=torch.float32):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)).to(dtype)

if len(sys.argv) > 1 and sys.argv[1] == "diff":
    for cin, sp in ((64, (8, 8, 32)), (32, (6, 5, 17)), (64, (37, 21, 45))):
        n = 2
        x = (rnd((n, *sp, cin), 191) * 1.2 + 0.1).to(dev)
        w = (rnd((1, cin, 3, 3, 3), 192) / math.sqrt(cin * 27)).to(dev)
        b = (rnd((1,), 193) * 0.1).to(dev)
        scale, shift = (rnd((n, cin), 194) * 0.2 + 1.0).to(dev), (rnd((n, cin), 195) * 0.1).to(dev)
        for pro in (False, True):
            kw = dict(kernel=3, padding=1, pre=(scale, shift) if pro else None, pre_act="silu" if pro else "none")
            a = ops.conv(x, w, b, force_cfg=20, **kw)
            t = ops.conv(x, w, b, force_cfg=13, **kw)
            d = (a - t).abs()
            nz = (d > 0).nonzero()
            print(f"cin{cin} {sp} prologue={pro}: mismatches {nz.shape[0]} of {d.numel()}, max|diff| {d.max().item():.3e}, |out|max {t.abs().max().item():.3g}; "
                  f"first at {nz[:6].tolist()}")
    sys.exit(0)

dtype = torch.bfloat16
x = rnd((1, 128, 128, 128, 64), 1).to(dev, dtype)
w = (rnd((1, 64, 3, 3, 3), 2) / 40).to(dev, dtype)
b = rnd((1,), 3).to(dev)
scale, shift = (rnd((1, 64), 4) * 0.2 + 1.0).to(dev), (rnd((1, 64), 5) * 0.1).to(dev)
for flags in ([0, 1, 2, 4, 3, 7] if os.environ.get("GM_CONV_COUT1_LTD") is None else [0]):
  ops._CONV_DEBUG_FLAGS = flags
  for cfg in ((13, 20) if flags == 0 else (20,)):
    for pro in ((True, False) if flags == 0 else (True,)):
        kw = dict(kernel=3, padding=1, pre=(scale, shift) if pro else None, pre_act="silu" if pro else "none", force_cfg=cfg)
        ops.conv(x, w, b, **kw); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv(x, w, b, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"cfg{cfg} ablate={flags} (1: no phase 1, 2: no phase 2, 4: no DMA) ltd={os.environ.get('GM_CONV_COUT1_LTD', 'auto')} prologue={pro}: {ms:.4f} ms = {0.2727 / ms:.2f} TB/s of the 272.7 MB a 64 -> 1 head at 128^3 moves")
