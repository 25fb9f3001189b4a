"""GPU: the energy-metered go / no-go on the tap loop of the dominant convolution (VERDICT r5 "next" 4).  Every row is ONE real launch of the library's
LDS-DMA kernels -- its real LDS image, DMA ring and barriers -- looped for >= 1.5 s with the package power sampled every 50 ms (bench.PowerSampler):
ms / launch, mean W, J / launch, pJ / FLOP.  All A/Bs quote joules, not milliseconds: the convolutions run at the package power cap, where
time = joules / cap and the same launch reads +-15 % in ms minutes apart.

  part A  the MFMA body, with the whole operand movement around it (python tools/taploop_energy.py bodies; the product library AS OF COMMIT 7c80f60: cfg 21 / 22
          left it with this measurement -- experiments/conv_mw, conv_w8 -- and now print an error row):
          (i)  cfg 14: v_mfma_f32_16x16x32_bf16, 256-voxel tile, 64-byte rows, 4 waves x 64 voxels, two work-groups per CU  (today's default)
          (ii) cfg 21: v_mfma_f32_32x32x16_bf16 on the SAME 256-voxel tile in 16-channel half-chunks (32-byte rows), three work-groups per CU
               cfg 22: v_mfma_f32_32x32x16_bf16 on a 512-voxel tile, 64-byte patch rows, 16-channel weight panels, two work-groups per CU
          on the C2 shapes 64->64 / 128->64 / 192->64 @128^3 and 128->128 / 384->128 @64^3.  GO only if a 32x32x16 body reaches <= 0.90 pJ/FLOP.
  part B  where the per-tile fixed energy of 64 -> 64 @128^3 sits (python tools/taploop_energy.py tile; GM_NATIVE_LIB = the `ablate` build, whose
          debug flags remove one piece at a time -- results are garbage, the instruction stream of everything else is unchanged):
          full (residual + statistics) | no statistics | no residual | no epilogue at all | no patch traffic | no weight traffic | tap loop alone
          and the same for 128 -> 64 (twice the taps per tile, the same fixed work).

Writes gpurun_out/taploop_energy_<mode>.json and prints one line per row."""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import PowerSampler  # noqa: E402
from generativemodels_amd import ops  # noqa: E402

dev = "cuda"
MIN_LOOP_S = float(os.environ.get("GM_TAPLOOP_SECONDS", "1.6"))


def meter(name, fn, flops, extra=None):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    n = max(20, int(MIN_LOOP_S * 1e3 / ms))
    for _ in range(n // 3):  # the package reaches its steady power state before sampling starts
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
    row = dict(row=name, launches=n, loop_s=round(ms * n / 1e3, 2), ms_per_launch=round(ms, 4), tflops=round(flops / ms / 1e9, 1),
               mean_w=None if not pw else pw["mean_w"], joules_per_launch=None if not pw else round(pw["mean_w"] * ms * 1e-3, 4),
               pj_per_flop=None if not pw else round(pw["mean_w"] * ms * 1e-3 / flops * 1e12, 4),
               sclk_mhz=None if not pw else (pw.get("sclk_mhz") or {}).get("mean"))
    if extra:
        row.update(extra)
    print(json.dumps(row), flush=True)
    return row


def operands(cin, cout, edge, with_res):
    x = torch.randn((1, edge, edge, edge, cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((cout, cin, 3, 3, 3), device=dev) / math.sqrt(cin * 27)).to(torch.bfloat16)
    b = torch.randn((cout,), device=dev)
    res = torch.randn((1, edge, edge, edge, cout), device=dev).to(torch.bfloat16) if with_res else None
    return x, w, b, res


SHAPES = [(64, 64, 128), (128, 64, 128), (192, 64, 128), (128, 128, 64), (384, 128, 64)]


def bodies():
    rows = []
    for cin, cout, edge in SHAPES:
        x, w, b, _ = operands(cin, cout, edge, False)
        flops = 2.0 * edge ** 3 * cin * cout * 27
        for cfg, body in ((14, "16x16x32, 256-voxel tile (default)"), (21, "32x32x16, 256-voxel tile, half-chunks"), (22, "32x32x16, 512-voxel tile")):
            try:
                rows.append(meter(f"{cin}->{cout} @{edge}^3 cfg {cfg}: {body}", lambda: ops.conv(x, w, b, kernel=3, padding=1, want_stats=True, force_cfg=cfg, ksplit=1),
                                  flops, dict(cfg=cfg, cin=cin, cout=cout, edge=edge)))
            except Exception as ex:  # a shape a configuration does not serve
                print(json.dumps(dict(row=f"{cin}->{cout} @{edge}^3 cfg {cfg}", error=str(ex)[:200])), flush=True)
        del x, w, b
    return rows


def tile():
    rows = []
    for cin, cout, edge in ((64, 64, 128), (128, 64, 128)):
        x, w, b, res = operands(cin, cout, edge, True)
        flops = 2.0 * edge ** 3 * cin * cout * 27

        def run(flags, want_stats, with_res):
            def fn():
                ops._CONV_DEBUG_FLAGS = flags
                try:
                    ops.conv(x, w, b, kernel=3, padding=1, want_stats=want_stats, res=res if with_res else None, force_cfg=14, ksplit=1)
                finally:
                    ops._CONV_DEBUG_FLAGS = 0
            return fn

        for name, flags, st, rs in (("full: residual + statistics", 0, True, True), ("no statistics", 0, False, True), ("no residual", 0, True, False),
                                    ("no residual, no statistics", 0, False, False), ("no epilogue at all (flag 256)", 256, False, False),
                                    ("no patch traffic (flag 1024)", 1024, True, True), ("no weight traffic (flag 512)", 512, True, True),
                                    ("no operand traffic (1024 + 512)", 1536, True, True), ("tap loop alone: no traffic, no epilogue (1792)", 1792, False, False)):
            rows.append(meter(f"{cin}->{cout} @{edge}^3 cfg 14, {name}", run(flags, st, rs), flops, dict(cfg=14, cin=cin, cout=cout, edge=edge, flags=flags)))
        del x, w, b, res
    return rows


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "bodies"
    idle = PowerSampler(0)
    idle.start()
    time.sleep(1.0)
    idle_w = idle.stop()
    print(json.dumps(dict(idle=idle_w, native_lib=os.environ.get("GM_NATIVE_LIB", "product library"))), flush=True)
    rows = bodies() if mode == "bodies" else tile()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(dict(mode=mode, idle=idle_w, native_lib=os.environ.get("GM_NATIVE_LIB", "product library"), rows=rows),
              open(os.path.join(ROOT, "gpurun_out", f"taploop_energy_{mode}.json"), "w"), indent=1)
