"""GPU: per-kernel effective shader clock and joules per launch for the kernels that own the C2 forward.

  python tools/clock_energy.py energy            each shape looped ~1.2 s with the package power sampled on a host thread (bench.PowerSampler):
                                                 ms / launch, mean W, J / launch, pJ / FLOP  -> gpurun_out/clock_energy_energy.json
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace ... -- python tools/clock_energy.py pmc
                                                 a few launches per shape under the counter pass
  python tools/clock_energy.py parse DIR         effective clock per kernel = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (the guide's recipe,
                                                 MI355X_MICROARCH.md "DVFS give-back") -> merged into profiles-ready JSON on stdout
Operands are random normal (switching power is data dependent: zero operands clock 20-30 % higher)."""
import csv
import glob
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONV = [("conv 64->64 @128^3 (+res, stats)", 64, 64, 128, True), ("conv 128->64 @128^3", 128, 64, 128, False), ("conv 192->64 @128^3", 192, 64, 128, False),
        ("conv 128->128 @64^3 (+res, stats)", 128, 128, 64, True), ("conv 384->128 @64^3", 384, 128, 64, False)]


def workloads():
    import torch
    from generativemodels_amd import ops

    dev = "cuda"
    out = []
    for name, cin, cout, edge, with_res in CONV:
        x = torch.randn((1, edge, edge, edge, cin), device=dev).to(torch.bfloat16)
        w = (torch.randn((cout, cin, 3, 3, 3), device=dev) / math.sqrt(cin * 27)).to(torch.bfloat16)
        b = torch.randn((cout,), device=dev)
        res = torch.randn((1, edge, edge, edge, cout), device=dev).to(torch.bfloat16) if with_res else None
        out.append((name, lambda x=x, w=w, b=b, res=res: ops.conv(x, w, b, kernel=3, padding=1, res=res, want_stats=True, force_cfg=14),
                    2.0 * edge ** 3 * cin * cout * 27, "conv_dma_kernel"))
    x = torch.randn((1, 128, 128, 128, 64), device=dev).to(torch.bfloat16)
    sc, sh = torch.rand((1, 64), device=dev) + 0.5, torch.randn((1, 64), device=dev) * 0.1
    y = torch.empty_like(x)
    out.append(("gn_apply 64 ch @128^3 (GroupNorm affine + SiLU)", lambda: ops.gn_apply(x, sc, sh, "silu", out=y), 0.0, "gn_apply"))
    L, c = 32768, 256
    qkv = torch.randn((1, L, 3 * c), device=dev).bfloat16()
    r = torch.randn((1, L, c), device=dev).bfloat16()
    out.append(("attention 32768 tokens x 256 (1 head)", lambda: ops.attention(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], 1, 1 / 16.0, res=r),
                4.0 * L * L * c, "attn_dma_kernel"))
    return out


def energy():
    import torch
    from bench import PowerSampler

    rows = []
    idle = PowerSampler(0)
    idle.start()
    time.sleep(1.0)
    idle_w = idle.stop()
    for name, fn, flops, _ in workloads():
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        n = max(20, int(1200.0 / ms))
        for _ in range(n // 4):  # bring the package to its steady power state before sampling
            fn()
        torch.cuda.synchronize()
        ps = PowerSampler(0)
        ps.start()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        pw = ps.stop()
        ms = e0.elapsed_time(e1) / n
        row = dict(kernel=name, launches=n, ms_per_launch=round(ms, 4), tflops=round(flops / ms / 1e9, 1) if flops else None,
                   mean_w=None if not pw else pw["mean_w"], max_w=None if not pw else pw.get("max_w"),
                   joules_per_launch=None if not pw else round(pw["mean_w"] * ms * 1e-3, 4),
                   pj_per_flop=None if not (pw and flops) else round(pw["mean_w"] * ms * 1e-3 / flops * 1e12, 3))
        rows.append(row)
        print(json.dumps(row), flush=True)
    doc = dict(idle=idle_w, rows=rows, note="package power (rocm-smi socket power) sampled every 50 ms while the one launch is looped; J / launch = mean W x ms")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(doc, open(os.path.join(ROOT, "gpurun_out", "clock_energy_energy.json"), "w"), indent=1)


def pmc():
    import torch

    for name, fn, _, _ in workloads():
        for _ in range(12):  # the first launches of a shape run while the clock is still settling: parse() takes the last ones
            fn()
        torch.cuda.synchronize()


def parse(d):
    rows = {}
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    trace = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            trace[r.get("Dispatch_Id")] = (float(r["Start_Timestamp"]), float(r["End_Timestamp"]))
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
                continue
            if "Start_Timestamp" in r and r["Start_Timestamp"]:
                t0, t1 = float(r["Start_Timestamp"]), float(r["End_Timestamp"])
            elif r.get("Dispatch_Id") in trace:
                t0, t1 = trace[r["Dispatch_Id"]]
            else:
                continue
            key = (r["Kernel_Name"].split("(")[0][:90], r.get("Grid_Size"), r.get("LDS_Block_Size"))
            rows.setdefault(key, []).append((float(r["Counter_Value"]), t1 - t0))
    out = []
    for key, v in rows.items():
        if not any(k in key[0] for k in ("conv_dma_kernel", "gn_apply", "attn_dma_kernel")):
            continue
        v = v[len(v) // 2:]  # steady state
        cyc = sum(a for a, _ in v) / len(v)
        ns = sum(b for _, b in v) / len(v)
        out.append(dict(kernel=key[0], grid=key[1], lds=key[2], dispatches=len(v), grbm_gui_active=round(cyc), duration_us=round(ns / 1e3, 2),
                        effective_clock_ghz=round(cyc / 8 / ns, 3)))
    out.sort(key=lambda r: (r["kernel"], -r["duration_us"]))
    print(json.dumps(dict(note="effective shader clock = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / dispatch duration; counter pass, so ~2-3 % "
                               "below the un-profiled clock (guide)", rows=out), indent=1))


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "energy"
    if mode == "energy":
        energy()
    elif mode == "pmc":
        pmc()
    else:
        parse(sys.argv[2])
