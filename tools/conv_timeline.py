"""GPU, bench-only build (python -m generativemodels_amd._build --variant timeline; run with GM_NATIVE_LIB=.../lib/libgmamd_timeline.so): where the cycles of one
LDS-DMA convolution tile go.  Thread 0 of every work-group stamps the shader clock at phase boundaries (conv_dma.hip TL_STAMP); this
script launches one convolution per shape / configuration and prints the median duration of every phase over the work-groups, split
over the work-groups (a persistent work-group stamps its second tile: the steady state).   usage: python tools/conv_timeline.py"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import torch
from generativemodels_amd import ops

dev = "cuda"


def run(cin, cout, size, cfg, with_res=True):
    x = torch.randn((1, size, size, size, cin), device=dev).bfloat16()
    w = (torch.randn((cout, cin, 3, 3, 3), device=dev) / math.sqrt(cin * 27)).bfloat16()
    if os.environ.get("GM_TL_ZERO"):  # all-zero operands: the same instruction stream at a lower switching power (DVFS check)
        x.zero_(); w.zero_()
    b = torch.randn((cout,), device=dev)
    res = torch.randn((1, size, size, size, cout), device=dev).bfloat16() if with_res else None
    kw = dict(kernel=3, padding=1, force_cfg=cfg, want_stats=True, res=res, ksplit=1)
    ops.conv(x, w, b, **kw)
    bm = 512 if cfg in (16, 18, 19, 22) else 256
    bn = 128 if cfg == 19 else 64
    nwg = (size ** 3 // bm) * ((cout + bn - 1) // bn)
    buf = torch.zeros((nwg, 64), dtype=torch.int64, device=dev)
    # smuggle the timeline buffer through GmConvDesc.kpartial (ksplit stays 0): patch ops.conv's descriptor via the debug hook
    ops._CONV_DEBUG_FLAGS = 4096 | int(os.environ.get("GM_TL_FLAGS", "0"))  # + 512: no weight traffic, + 1024: no patch traffic (timeline_ablate build)
    ops._CONV_TIMELINE_BUFFER = buf
    try:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv(x, w, b, **kw)
        e1.record()
        torch.cuda.synchronize()
        global LAST_MS
        LAST_MS = e0.elapsed_time(e1)
    finally:
        ops._CONV_DEBUG_FLAGS = 0
        ops._CONV_TIMELINE_BUFFER = None
    t = buf.cpu().numpy().astype("int64")
    return t, nwg


LAST_MS = 0.0


def report(name, t, nchunks, gpc=9, stride=10):
    """gpc = tap groups per stamped chunk, stride = stamp slots per chunk (cfg 22: 18 groups = 2 halves x 9, chunks 0-1 stamped)"""
    import numpy as np
    ok = (t[:, 63] > 0) & (t[:, 0] > 0)
    # s_memtime counters are PER XCD (work-group b runs on XCD b % 8) and not synchronised with each other: a span is only meaningful inside one XCD.
    # (Round 4 printed max - min over the whole chip: "6080423.750 GHz".)  The tick is not the shader clock either: 16 tile rounds x 44.3 k ticks in
    # 0.458 ms = 1.55 G ticks/s while GRBM_GUI_ACTIVE / 8 / duration gives 1.97 GHz for the same launch (profiles/r05_clock_energy.json) -- read the
    # phase lengths below as RELATIVE shares of a tile's life; the effective clock comes from the counter pass (tools/clock_energy.py).
    xcd = np.arange(len(t)) % 8
    spans = [int(t[ok & (xcd == x), 63].max() - t[ok & (xcd == x), 0].min()) for x in range(8) if (ok & (xcd == x)).any()]
    t = t[ok]
    order = np.argsort(t[:, 0])
    t = t[order]
    span = sorted(spans)[len(spans) // 2] if spans else 0
    print(f"--- {name}: {len(t)} work-groups, kernel span (median over the XCDs' own counters) {span} s_memtime ticks in {LAST_MS:.3f} ms = "
          f"{span / max(LAST_MS, 1e-9) / 1e6:.3f} G ticks/s (stamped launch; not the shader clock: see tools/clock_energy.py)")
    for label, sel in (("all stamped work-groups", slice(0, None)),):
        tt = t[sel]
        if len(tt) == 0:
            continue
        med = lambda a: int(np.median(a))  # noqa: E731
        rows = [("tile top -> first panels + addend loads issued", tt[:, 55] - tt[:, 0]),
                ("accumulator init, addend -> LDS", tt[:, 56] - tt[:, 55]),
                ("wait for the first DMAs (+ barrier)", tt[:, 2] - tt[:, 56])]
        prev = tt[:, 2]
        nst = min(nchunks, 5 if gpc == 9 else 2)
        for c in range(nst):
            groups = []
            for g in range(gpc):
                cur = tt[:, 3 + c * stride + g]
                groups.append(cur - prev)
                prev = cur
            g_first = np.stack(groups[:gpc - 1], 1)
            rows.append((f"chunk {c}: tap groups 0-{gpc - 2} (3 taps), median of per-group medians", np.median(g_first, 1)))
            rows.append((f"chunk {c}: tap group {gpc - 1}" + (" + chunk boundary (patch reload)" if c + 1 < nchunks else " (last)"), groups[gpc - 1]))
        if nchunks <= nst:
            rows.append(("main loop end -> residual requests, shortcut", tt[:, 61] - tt[:, 60]))
            rows.append(("barrier, next tile decoded + its patch requested", tt[:, 57] - tt[:, 61]))
            rows.append(("epilogue: LDS transpose, residual, stores", tt[:, 62] - tt[:, 57]))
            rows.append(("statistics reduce + store", tt[:, 63] - tt[:, 62]))
        rows.append(("tile top -> tile end (the stamped tile: the second of a persistent work-group)", tt[:, 63] - tt[:, 0]))
        print(f"  [{label}]")
        for lab, a in rows:
            print(f"    {lab:78s} {med(a):8d} cycles")


SHAPES = [(64, 64, 128, 11), (64, 64, 128, 18), (192, 64, 128, 11), (384, 128, 64, 11), (256, 256, 32, 11)]
if os.environ.get("GM_TL_SHAPES"):  # e.g. "192,64,128,11;384,128,64,11"
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["GM_TL_SHAPES"].split(";")]
for cin, cout, size, cfg in SHAPES:
    t, nwg = run(cin, cout, size, cfg)
    if cfg == 22:
        report(f"{cin}->{cout} @ {size}^3 cfg{cfg}", t, cin // 32, gpc=18, stride=18)
    else:
        report(f"{cin}->{cout} @ {size}^3 cfg{cfg}", t, cin // (16 if cfg == 21 else 32))  # cfg 21 advances K in 16-channel half-chunks
