"""bench.py -- headline benchmark of the MI355X sampling path (BASELINE.json: 3D volumes/sec, DDIM 50-step sample of
1x128^3 volumes; UNet forward ms/step), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]            # N = 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (config C2): DiffusionModelUNet(3, 1, 1, num_channels=(64,128,256), 2 res blocks, attention in the
mid block with one 256-wide head) in bf16, DDIMScheduler(1000, scaled_linear_beta 0.0005..0.0195, clip_sample=False),
set_timesteps(50), one 1x1x128^3 volume per GPU per step.  A "step" = one complete 50-step sampling chain (50 UNet forwards +
50 fused scheduler steps) with the noise already resident in HBM.  Random-init weights (all-zero parameters re-randomised so
the network is not the zero function), synthetic Gaussian noise input.  Volumes are independent chains: ranks share nothing
on the data path (weak scaling, no collective); the barrier / all-reduce below only brackets the timing.

Prints ONE JSON line on rank 0 with the driver's contract fields plus
  "roofline":     the dominant kernel (implicit-GEMM convolution) -- algorithmic FLOPs per launch / HIP-event time per launch
                  over one instrumented forward, against the dense bf16 MFMA peak;
  "cpu_baseline": the CPU oracle (oracle/restatement.py, the reference algorithm restated on torch-CPU fp32) timed on this
                  box's host cores on a bounded sample (one of the 50 steps at full size), extrapolated to volumes/s."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0

C2 = dict(spatial_dims=3, in_channels=1, out_channels=1, num_channels=(64, 128, 256), attention_levels=(False, False, False),
          num_res_blocks=2, num_head_channels=(0, 0, 256), norm_num_groups=32)


def rerandomize_zero_params(state_dict, seed=1234, std=0.05):
    """A freshly constructed UNet is the zero function (zero-initialised conv2 / out conv): give every all-zero parameter
    N(0, std) values from a fixed seed so the benchmark does real arithmetic on non-degenerate data."""
    g = torch.Generator().manual_seed(seed)
    for _, p in state_dict.items():
        if p.is_floating_point() and p.numel() > 0 and p.abs().max() == 0:
            p.copy_(torch.randn(p.shape, generator=g) * std)
    return state_dict


# conv_fast_kernel<T, WM, WN, MF, MINW> instantiation behind each fast cfg (csrc/conv_fast.hip dispatch_fast)
FAST_CFG_TEMPLATE = {5: "4, 1, 4, 2", 6: "4, 2, 4, 2", 7: "8, 1, 4, 2", 8: "8, 1, 2, 4", 9: "8, 2, 2, 4", 10: "16, 1, 2, 4"}
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic_per_kernel.json")


def measured_hbm_traffic(label: str, dtype) -> dict:
    """HBM bytes per launch of the roofline kernel from the committed PMC passes (tools/pmc_traffic.sh: rocprofv3 --pmc
    FETCH_SIZE and --pmc WRITE_SIZE in separate runs of this same bench command; FETCH_SIZE doubled per the gfx950 note in
    MI355X_MICROARCH.md, WRITE_SIZE calibrated 1:1 on the copy kernel).  Counters cannot be collected inside the timed run,
    so the figure is read from profiles/; null when no committed pass covers this kernel."""
    try:
        cfg = int(label.split("cfg")[1].rstrip(">"))
        rows = json.load(open(TRAFFIC_FILE))
    except Exception:
        return dict(traffic=None)
    elem = "unsigned short" if dtype == torch.bfloat16 else "float"
    # template arguments <T, waves, fragments, stride, min waves / EU[, kernel extent]>: the trailing kernel-extent argument (3, or 2 for the
    # sub-pixel up-sampling variant) was added after the first PMC files were recorded, so cfg 11 matches "...4>" and "...4, 3>"
    sym = (f"conv_fast_kernel<{elem}, {FAST_CFG_TEMPLATE[cfg]}>" if cfg in FAST_CFG_TEMPLATE else
           f"conv_dma_kernel<{elem}, 8, 2, 1, 4" if cfg == 11 else f"conv_dma_kernel<{elem}, 4, 4, 1, 2" if cfg == 14 else
           f"conv_dma_kernel<{elem}, 8, 1, 2, 2" if cfg == 15 else None)
    for r in rows:
        if sym is not None and sym in r["kernel"] and (cfg in FAST_CFG_TEMPLATE or r["kernel"].split(sym)[1][:4] in (">(Gm", ", 3>")):
            return dict(traffic=round(r["hbm_mb_per_launch_corrected"] * 1e6), traffic_unit="bytes/launch (avg over the same launches)",
                        traffic_source=os.path.relpath(TRAFFIC_FILE, ROOT), traffic_launches_sampled=r["launches"])
    return dict(traffic=None)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed sampling chains (volumes) per GPU")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=128, help="volume edge (128 = the BASELINE config)")
    ap.add_argument("--inference-steps", type=int, default=50)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--graph", type=int, default=0, help="replay the UNet forward from a HIP graph (measured slower than eager launches on ROCm 7.2: 32.9 vs 27.7 ms per iteration)")
    ap.add_argument("--cpu-baseline", default="full", choices=["full", "small", "off"])
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: generativemodels_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # RCCL over xGMI; used only to bracket the timing
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from generativemodels_amd import ops
    from generativemodels_amd.inferers import DiffusionInferer
    from generativemodels_amd.networks.nets import DiffusionModelUNet
    from generativemodels_amd.networks.schedulers import DDIMScheduler

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    model = DiffusionModelUNet(**C2).eval()
    sd = rerandomize_zero_params({k: v.clone() for k, v in model.state_dict().items()})
    model.load_state_dict(sd)
    model = model.to(dev, dtype)
    sched = DDIMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0005, beta_end=0.0195, clip_sample=False)
    sched.set_timesteps(args.inference_steps)
    inferer = DiffusionInferer(sched, use_hip_graph=bool(args.graph))
    shape = (1, 1, args.size, args.size, args.size)
    noise = torch.randn(shape, generator=torch.Generator().manual_seed(7 + rank)).to(dev, dtype)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    for _ in range(args.warmup):
        out = inferer.sample(noise, model, sched, verbose=False)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = inferer.sample(noise, model, sched, verbose=False)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    finite = bool(torch.isfinite(out.float()).all().item())

    # ---- per-kernel roofline: one instrumented eager forward (HIP events on the launch stream around every launch) ---------
    roof = None
    fwd_ms = None
    breakdown = {}
    if rank == 0:
        tt = torch.tensor([500.0], device=dev)
        model(noise, tt)  # warm
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            model(noise, tt)
        e1.record()
        torch.cuda.synchronize()
        fwd_ms = e0.elapsed_time(e1) / 3
        ops.start_profile()
        model(noise, tt)
        rec = ops.stop_profile()
        for name, meta, ms in rec:
            b = breakdown.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            b["launches"] += 1
            b["ms"] += ms
            b["flops"] += meta["flops"]
            b["bytes"] += meta["bytes"]
        conv = {k: v for k, v in breakdown.items() if k.startswith("conv_igemm")}
        if conv:
            name, b = max(conv.items(), key=lambda kv: kv[1]["ms"])
            achieved = b["flops"] / (b["ms"] * 1e-3) / 1e12
            roof = dict(kernel=name, bound="mfma", achieved=round(achieved, 2), peak=MFMA_BF16_PEAK_TFLOPS if dtype == torch.bfloat16 else 157.3,
                        unit="TFLOP/s", frac=round(achieved / (MFMA_BF16_PEAK_TFLOPS if dtype == torch.bfloat16 else 157.3), 4), traffic=None,
                        launches=b["launches"], avg_launch_ms=round(b["ms"] / b["launches"], 4),
                        algorithmic_gflop_per_launch=round(b["flops"] / b["launches"] / 1e9, 2),
                        algorithmic_mb_per_launch=round(b["bytes"] / b["launches"] / 1e6, 2))

        if roof is not None:
            roof.update(measured_hbm_traffic(roof["kernel"], dtype))

    # ---- CPU baseline: the oracle on this box's host cores, bounded sample ----------------------------------------------------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_baseline != "off":
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import restatement as R  # test infrastructure: the CPU statement of the reference algorithm (checker / baseline only)

        cores = os.cpu_count() or 1
        torch.set_num_threads(cores)
        size = args.size if args.cpu_baseline == "full" else min(args.size, 48)
        x = torch.randn((1, 1, size, size, size), generator=torch.Generator().manual_seed(7))
        sd32 = {k: v.float() for k, v in sd.items()}
        with torch.no_grad():
            R.unet_forward(sd32, C2, x[..., : size // 2, : size // 2, : size // 2].contiguous(), torch.tensor([500.0]))  # warm the thread pool
            c0 = time.perf_counter()
            eps = R.unet_forward(sd32, C2, x, torch.tensor([500.0]))
            R.ddim_step(sched.alphas_cumprod, 1000, args.inference_steps, eps, 500, x, clip_sample=False)
            cpu_s = time.perf_counter() - c0
        scale = (args.size / size) ** 3
        cpu = dict(value=round(1.0 / (cpu_s * scale * args.inference_steps), 8), unit="volumes/s", cores=cores, kind="port",
                   seconds_per_step=round(cpu_s * scale, 3),
                   sample=f"1 of {args.inference_steps} DDIM steps (UNet forward + scheduler step, fp32, torch-CPU oracle) at 1x1x{size}^3"
                          + ("" if size == args.size else f", scaled x{scale:.0f} to {args.size}^3 by voxel count") + f", x{args.inference_steps} steps")

    if rank == 0:
        vol_s = world * args.steps / elapsed
        line = {
            "metric": "3D volumes/sec (DDIM 50-step sample, 1x128^3)", "value": round(vol_s, 5), "unit": "volumes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (Gaussian noise volumes, random-init weights)",
            "config": {"workload": f"C2: 3D DiffusionModelUNet(64,128,256; 2 res blocks; mid-block attention 1 head x 256) DDIM-{args.inference_steps} "
                                   f"sampling of 1x1x{args.size}^3 volumes, 1 volume per GPU per step",
                       "volumes_per_gpu_per_step": 1, "inference_steps": args.inference_steps,
                       "parallelism": f"{world} independent replicas (batch-sharded, no data-path collective)", "hip_graph": bool(args.graph)},
            "unet_forward_ms": None if fwd_ms is None else round(fwd_ms, 3),
            "ms_per_ddim_iteration": round(1e3 * elapsed / args.steps / args.inference_steps, 3),
            "output_finite": finite,
            "roofline": roof, "cpu_baseline": cpu,
            "speedup_vs_cpu": None if cpu is None else round(vol_s / cpu["value"], 1),
            "kernel_breakdown_ms": {k: dict(launches=v["launches"], ms=round(v["ms"], 3),
                                            tflops=round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2), gbs=round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1))
                                    for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1]["ms"])},
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
