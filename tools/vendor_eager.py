"""The "vendor eager" comparator of SURVEY.md 8(d): the reference algorithm (oracle/restatement.py = the reference's op sequence
on torch ops) placed on the GPU, i.e. PyTorch-ROCm eager with MIOpen / rocBLAS kernels, in the same dtype and at the same
config-C2 shape as bench.py.  This is what a user of the reference gets on an MI355X today; it is a baseline, never the product
path (the oracle is test infrastructure: this tool is part of bench.py's baseline leg).

    python tools/vendor_eager.py [--size 128] [--dtype bf16] [--iters 3] [--channels-last 0]

Prints one JSON line: ms per UNet forward, ms per DDIM iteration (forward + the reference's scheduler-step op sequence)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--channels-last", type=int, default=0)
    args = ap.parse_args()
    import restatement as R
    from bench import C2, rerandomize_zero_params
    from generativemodels_amd.networks.nets import DiffusionModelUNet  # parameter shapes / default init only
    from generativemodels_amd.networks.schedulers import DDIMScheduler

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    sd = rerandomize_zero_params({k: v.clone() for k, v in DiffusionModelUNet(**C2).state_dict().items()})
    sd = {k: v.to(dev, dtype) if v.is_floating_point() else v.to(dev) for k, v in sd.items()}
    if args.channels_last:
        sd = {k: (v.contiguous(memory_format=torch.channels_last_3d) if v.ndim == 5 else v) for k, v in sd.items()}
    sched = DDIMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0005, beta_end=0.0195, clip_sample=False)
    sched.set_timesteps(50)
    x = torch.randn((1, 1, args.size, args.size, args.size), generator=torch.Generator().manual_seed(7)).to(dev, dtype)
    if args.channels_last:
        x = x.contiguous(memory_format=torch.channels_last_3d)
    t = torch.tensor([500.0], device=dev)
    acp = sched.alphas_cumprod.to(dev)

    def iteration():
        eps = R.unet_forward(sd, C2, x, t)
        return R.ddim_step(acp, 1000, 50, eps, 500, x, clip_sample=False)

    with torch.no_grad(), torch.device(dev):
        w0 = time.perf_counter()
        iteration()  # MIOpen kernel selection happens here
        torch.cuda.synchronize()
        warm_s = time.perf_counter() - w0
        iteration()
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(args.iters):
            R.unet_forward(sd, C2, x, t)
        e1.record()
        for _ in range(args.iters):
            iteration()
        e2.record()
        torch.cuda.synchronize()
    fwd = e0.elapsed_time(e1) / args.iters
    it = e1.elapsed_time(e2) / args.iters
    print(json.dumps(dict(kind="vendor_eager", what="reference op sequence on PyTorch-ROCm eager (MIOpen/rocBLAS), same shape and dtype",
                          torch=torch.__version__, dtype=args.dtype, size=args.size, channels_last=bool(args.channels_last),
                          unet_forward_ms=round(fwd, 2), ms_per_ddim_iteration=round(it, 2), volumes_per_s=round(1e3 / (it * 50), 5),
                          first_call_s=round(warm_s, 1), peak_mem_gb=round(torch.cuda.max_memory_allocated() / 1e9, 1))), flush=True)


if __name__ == "__main__":
    main()
