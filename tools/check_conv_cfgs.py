"""GPU: correctness + timing of LDS-DMA convolution tile configurations against the reference configuration (cfg 11) on the C2 / C3 shapes,
with every fused feature the ResnetBlock uses (bias, timestep row, residual, fused 1x1 shortcut over one or two sources, output statistics,
folded nearest-2x input) and ragged extents.  usage: python tools/check_conv_cfgs.py [cfg,cfg,...] [--time-only]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from generativemodels_amd import ops

dev = "cuda"
cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else [18, 19]
REF = 11
g = torch.Generator(device=dev).manual_seed(3)


def rn(shape, scale=1.0, dtype=torch.bfloat16):
    return (torch.randn(shape, generator=g, device=dev) * scale).to(dtype)


def run(cfg, x, w, b, **kw):
    y = ops.conv(x, w, b, kernel=3, padding=1, force_cfg=cfg, **kw)
    st = getattr(y, "_gm_cstats", None)
    return y, (None if st is None else st.sum(0).clone())


CASES = [  # name, cin, cout, spatial, features
    ("64->64@48^3 res+stats+row", 64, 64, (48, 48, 48), dict(res=True, row=True, stats=True)),
    ("64->128@32^3 skip1+stats", 64, 128, (32, 32, 32), dict(skip=(64,), stats=True, row=True)),
    ("192->64@40x36x52 skip2+stats", 192, 64, (40, 36, 52), dict(skip=(128, 64), stats=True)),
    ("384->128@24^3 skip2", 384, 128, (24, 24, 24), dict(skip=(256, 128), stats=True)),
    ("128->256@16^3 skip1", 128, 256, (16, 16, 16), dict(skip=(128,), stats=True)),
    ("128->128 up 20^3->40^3", 128, 128, (20, 20, 20), dict(up=True, stats=True)),
    ("96->160@19x21x23 ragged", 96, 160, (19, 21, 23), dict(res=True, stats=True)),
    ("fp32 64->128@24^3 skip1", 64, 128, (24, 24, 24), dict(skip=(64,), stats=True, fp32=True)),
]
bad = 0
ops._CONV_DEBUG_FLAGS = int(os.environ.get("CONV_FLAGS", "0"))
if "--time-only" not in sys.argv:
    for name, cin, cout, sp, f in CASES:
        dt = torch.float32 if f.get("fp32") else torch.bfloat16
        x = rn((1, *sp, cin), dtype=dt)
        w = rn((cout, cin, 3, 3, 3), 1 / math.sqrt(cin * 27), dt)
        b = rn((cout,), 0.1, torch.float32)
        osp = tuple(2 * s for s in sp) if f.get("up") else sp
        kw = dict(want_stats=bool(f.get("stats")), upsample=bool(f.get("up")))
        if f.get("res"):
            kw["res"] = rn((1, *osp, cout), dtype=dt)
        if f.get("row"):
            kw["rowvec"] = rn((1, cout), 0.3, torch.float32)
        if f.get("skip"):
            parts = [rn((1, *osp, c), dtype=dt) for c in f["skip"]]
            sw = rn((cout, sum(f["skip"]), 1, 1, 1), 1 / math.sqrt(sum(f["skip"])), dt)
            kw["skip"] = (parts, sw, rn((cout,), 0.1, torch.float32))
        if f.get("up"):
            kw["allow_subpixel"] = False
        yr, sr = run(REF, x, w, b, **kw)
        torch.cuda.synchronize()
        for cfg in cfgs:
            try:
                y, s = run(cfg, x, w, b, **kw)
                torch.cuda.synchronize()
            except Exception as ex:
                print(f"{name:34s} cfg{cfg}: n/a ({str(ex)[:60]})")
                continue
            err = (y.float() - yr.float()).abs().max().item()
            scale = yr.float().abs().max().item()
            serr = 0.0 if s is None else ((s - sr).abs() / sr.abs().clamp_min(1.0)).max().item()
            ok = err <= (1e-5 if dt == torch.float32 else 2 ** -7) * scale and serr <= 1e-4 and bool(torch.isfinite(y.float()).all())
            bad += 0 if ok else 1
            print(f"{name:34s} cfg{cfg}: max|y - y_cfg{REF}| {err:.3e} (scale {scale:.3g})  stats rel err {serr:.2e}  {'ok' if ok else 'MISMATCH'}", flush=True)
    print("correctness:", "ALL OK" if bad == 0 else f"{bad} MISMATCHES")

TIMING = [  # name, cin, cout, spatial, upsample, skip sources
    ("64->64@128^3", 64, 64, (128, 128, 128), False, None),
    ("128->64@128^3 skip2", 128, 64, (128, 128, 128), False, (64, 64)),
    ("192->64@128^3", 192, 64, (128, 128, 128), False, None),
    ("64->128@64^3", 64, 128, (64, 64, 64), False, None),
    ("128->128@64^3", 128, 128, (64, 64, 64), False, None),
    ("384->128@64^3", 384, 128, (64, 64, 64), False, None),
    ("128->256@32^3", 128, 256, (32, 32, 32), False, None),
    ("256->256@32^3", 256, 256, (32, 32, 32), False, None),
    ("512->256@32^3", 512, 256, (32, 32, 32), False, None),
    ("64->64@256^3 (C3 decoder)", 64, 64, (256, 256, 256), False, None),
]
for name, cin, cout, sp, up, skip in TIMING:
    x = rn((1, *sp, cin))
    w = rn((cout, cin, 3, 3, 3), 1 / math.sqrt(cin * 27))
    b = rn((cout,), 0.1, torch.float32)
    kw = dict(want_stats=True, rowvec=rn((1, cout), 0.3, torch.float32))
    if skip:
        kw["skip"] = ([rn((1, *sp, c)) for c in skip], rn((cout, sum(skip), 1, 1, 1), 0.1), None)
    else:
        kw["res"] = rn((1, *sp, cout))
    flops = 2.0 * math.prod(sp) * cout * (cin * 27 + (sum(skip) if skip else 0))
    line = f"{name:28s}"
    for cfg in [REF] + cfgs:
        try:
            run(cfg, x, w, b, **kw)
            torch.cuda.synchronize()
            n = 10
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                ops.conv(x, w, b, kernel=3, padding=1, force_cfg=cfg, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            line += f" | cfg{cfg}: {ms:6.3f} ms {flops / ms / 1e9:6.0f} TF/s"
        except Exception as ex:
            line += f" | cfg{cfg}: n/a ({str(ex)[:30]})"
    print(line, flush=True)
    del x, w
sys.exit(1 if bad else 0)
