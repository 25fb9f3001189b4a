"""GPU micro-benchmark: the dominant C2 convolution shapes through ops.conv for selected tile configurations.
usage: python tools/bench_conv.py [cfg,cfg,...]   (prints TFLOP/s per shape and configuration)"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from generativemodels_amd import ops

dev = "cuda"
SHAPES = [  # name, Cin, Cout, spatial (input), upsample, prologue
    ("64->64@128^3 gn+silu", 64, 64, (128, 128, 128), False, True),
    ("192->64@128^3 gn+silu", 192, 64, (128, 128, 128), False, True),
    ("128->128 up 64^3->128^3", 128, 128, (64, 64, 64), True, False),
    ("128->128@64^3 gn+silu", 128, 128, (64, 64, 64), False, True),
    ("384->128@64^3 gn+silu", 384, 128, (64, 64, 64), False, True),
    ("256->256@32^3 gn+silu", 256, 256, (32, 32, 32), False, True),
    ("64->64@128^3 plain", 64, 64, (128, 128, 128), False, False),
]
cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [5, 6, 7, 8, 9]
dtype = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
ops._CONV_DEBUG_FLAGS = int(os.environ.get("CONV_FLAGS", "0"))
PLAIN = bool(int(os.environ.get("BENCH_PLAIN", "0")))  # drop the fused prologue from every shape (the LDS-DMA kernel's domain)
for name, cin, cout, sp, up, pro in SHAPES:
    if PLAIN:
        if not pro:
            continue
        name, pro = name.replace("gn+silu", "plain"), False
    x = torch.randn((1, *sp, cin), device=dev).to(dtype)
    w = (torch.randn((cout, cin, 3, 3, 3), device=dev) / math.sqrt(cin * 27)).to(dtype)
    b = torch.randn((cout,), device=dev)
    if os.environ.get("BENCH_ZERO"):  # all-zero operands: same instruction stream, lower switching power (DVFS check)
        x.zero_(); w.zero_()
    pre = (torch.rand((1, cin), device=dev) + 0.5, torch.randn((1, cin), device=dev) * 0.1) if pro else None
    osp = tuple(s * 2 for s in sp) if up else sp
    flops = 2.0 * math.prod(osp) * cin * cout * 27
    line = f"{name:28s}"
    ref = None
    for cfg in cfgs:
        try:
            kw = dict(kernel=3, padding=1, upsample=up, pre=pre, pre_act="silu" if pro else "none", force_cfg=cfg)
            y = ops.conv(x, w, b, **kw)
            torch.cuda.synchronize()
            if ref is None:
                ref = y.float()
            else:
                err = (y.float() - ref).abs().max().item()
                assert err < 0.1 or ops._CONV_DEBUG_FLAGS, (name, cfg, err)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.conv(x, w, b, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            line += f" | cfg{cfg}: {ms:7.3f} ms {flops / ms / 1e9:7.1f} TF/s"
        except Exception as ex:  # noqa
            line += f" | cfg{cfg}: n/a ({str(ex)[:40]})"
    print(line, flush=True)
    del x, w, y
