"""GPU: one training step of BASELINE config C4 on ONE MI355X (the per-rank work of the batch-sharded DDP job): a frozen AutoencoderKL
encodes B x 1 x S^3 volumes to B x 4 x (S/8)^3 latents, DiffusionInferer.__call__ noises them and runs the 41.7 M-parameter latent UNet
forward WITH gradients (native kernels in both directions), MSE loss, backward, GradientReducer.finish(), Adam.
dtype "mixed" (the default) is the reference's own arithmetic for this loop (ddpm_training_ddp.py:129,249-270: fp32 parameters, forward under
autocast, GradScaler): fp32 master parameters and fp32 gradients / Adam state, bf16 activations and MFMA operands inside
`generativemodels_amd.autocast(torch.bfloat16)`.  "bf16" casts the parameters themselves (narrower than the reference: an lr-sized Adam update
is below half a bf16 ulp of most weights), "fp32" runs the exact-fp32 MFMA kernels.
"graph" replays encode + forward + loss + backward from ONE HIP graph (generativemodels_amd.GraphedForwardBackward): the eager step is host-bound
(~620 launches at ~50 us of Python each against ~19 ms of kernels).
usage: python tools/bench_train.py [size=256] [batch=1] [dtype=mixed|bf16|fp32] [steps=3] [eager|graph]     (under torchrun: one process per GPU, RCCL)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import contextlib

import torch
import torch.nn.functional as F

import generativemodels_amd as gm
from bench import rerandomize_zero_params
from generativemodels_amd import ops
from generativemodels_amd.inferers import LatentDiffusionInferer
from generativemodels_amd.networks.nets import AutoencoderKL, DiffusionModelUNet
from generativemodels_amd.networks.schedulers import DDPMScheduler
from generativemodels_amd.parallel import GradientReducer

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mode = sys.argv[3] if len(sys.argv) > 3 else "mixed"
if mode not in ("mixed", "bf16", "fp32"):
    raise SystemExit("dtype must be mixed, bf16 or fp32")
dt = torch.bfloat16 if mode == "bf16" else torch.float32          # dtype of the parameters (and of the optimizer state)
region = (lambda: gm.autocast(torch.bfloat16)) if mode == "mixed" else contextlib.nullcontext
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
use_graph = len(sys.argv) > 5 and sys.argv[5] == "graph"
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
force_reducer = os.environ.get("GM_FORCE_REDUCER", "0") == "1"  # one rank, but the whole RCCL exchange path (side stream, buckets) runs
if world > 1 or force_reducer:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
dev = f"cuda:{local}"
torch.manual_seed(0)
ae = AutoencoderKL(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=4, num_channels=(64, 128, 128, 128), num_res_blocks=2,
                   attention_levels=(False, False, False, False), with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False).eval().to(dev, dt)
for p in ae.parameters():
    p.requires_grad_(False)
unet = DiffusionModelUNet(spatial_dims=3, in_channels=4, out_channels=4, num_channels=(64, 128, 256), attention_levels=(False, True, True),
                          num_res_blocks=2, num_head_channels=(0, 128, 256))
unet.load_state_dict(rerandomize_zero_params({k: v.clone() for k, v in unet.state_dict().items()}))
unet = unet.to(dev, dt)
sched = DDPMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0205)
inf = LatentDiffusionInferer(sched, scale_factor=1.0)
opt = torch.optim.Adam(unet.parameters(), lr=1e-5)
red = GradientReducer(unet.parameters(), force=force_reducer, usage_check_every=int(os.environ.get("GM_REDUCER_USAGE_EVERY", "1")),
                      static_graph=os.environ.get("GM_REDUCER_STATIC", "0") == "1")
if os.environ.get("GM_REDUCER_NO_COLLECTIVE", "0") == "1":  # diagnosis only: the bucket / hook machinery without the RCCL launches (what part of the
    class _Done:                                              # one-rank tax is the collective kernels sharing the GPU with backward)
        def wait(self):
            pass
    red._launch = lambda i: red._work.__setitem__(i, _Done())
g = torch.Generator().manual_seed(100 + rank)
imgs = torch.randn((batch, 1, size, size, size), generator=g).to(dev, dt)
lat = size // 8


def loss_fn(images, noise, t):
    with region():
        pred = inf(inputs=images, autoencoder_model=ae, diffusion_model=unet, noise=noise, timesteps=t)
    return F.mse_loss(pred.float(), noise.float())


graphed = None
if use_graph:
    graphed = gm.GraphedForwardBackward(loss_fn, (imgs, torch.randn((batch, 4, lat, lat, lat), device=dev, dtype=dt),
                                                  torch.randint(0, 1000, (batch,), device=dev)), unet.parameters(), reducer=red)


def step():
    noise = torch.randn((batch, 4, lat, lat, lat), generator=g).to(dev, dt)
    t = torch.randint(0, 1000, (batch,), generator=g).to(dev)
    if graphed is not None:
        loss = graphed(imgs, noise, t)
    else:
        red.zero_grad() if red.active else opt.zero_grad(set_to_none=True)  # one fill per bucket: .grad stays a view of its flat bucket
        loss = loss_fn(imgs, noise, t)
        loss.backward()
    red.finish()
    opt.step()
    return loss


def phase_times():
    """One instrumented step: encode / forward / backward / optimizer, each bracketed by synchronize."""
    out = {}
    noise = torch.randn((batch, 4, lat, lat, lat), generator=g).to(dev, dt)
    t = torch.randint(0, 1000, (batch,), generator=g).to(dev)
    opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad(), region():
        z = ae.encode_stage_2_inputs(imgs)
    torch.cuda.synchronize(); out["encode_ms"] = (time.perf_counter() - t0) * 1e3; t0 = time.perf_counter()
    noisy = sched.add_noise(z, noise, t)
    with region():
        pred = unet.forward_train(noisy, t)
    loss = F.mse_loss(pred.float(), noise.float())
    torch.cuda.synchronize(); out["forward_ms"] = (time.perf_counter() - t0) * 1e3; t0 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); out["backward_ms"] = (time.perf_counter() - t0) * 1e3; t0 = time.perf_counter()
    red.finish()
    opt.step()
    torch.cuda.synchronize(); out["reduce_optimizer_ms"] = (time.perf_counter() - t0) * 1e3
    return {k: round(v, 2) for k, v in out.items()}


losses = [float(step()) for _ in range(2)]  # warm-up (weight packing, allocator)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
t0 = time.perf_counter()
for _ in range(steps):
    losses.append(float(step()))
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
dt_step = (time.perf_counter() - t0) / steps
phases = phase_times()
ops.start_profile()
noise = torch.randn((batch, 4, lat, lat, lat), generator=g).to(dev, dt)
tt = torch.randint(0, 1000, (batch,), generator=g).to(dev)
with region():
    pred = unet.forward_train(sched.add_noise(torch.randn_like(noise), noise, tt), tt)
F.mse_loss(pred.float(), noise.float()).backward()
agg = {}
for name, meta, ms in ops.stop_profile():
    a = agg.setdefault(name, dict(launches=0, ms=0.0, flops=0.0))
    a["launches"] += 1; a["ms"] += ms; a["flops"] += meta["flops"]
if rank == 0:
    print(json.dumps(dict(config=f"C4 per-rank training step: {batch} x 1 x {size}^3 volumes -> {batch} x 4 x {lat}^3 latents, 41.7 M-parameter UNet",
                          dtype=("mixed: fp32 parameters / gradients / Adam state, bf16 compute" if mode == "mixed" else str(dt).split(".")[-1]),
                          gradient_bucket_bytes=sum(p.numel() * p.element_size() for p in unet.parameters()), n_gpus=world, step_ms=round(dt_step * 1e3, 2), hip_graph=use_graph,
                          gradient_exchange=(f"RCCL all-reduce, {len(red.buckets)} buckets, {red.launched_in_backward} launched during backward "
                                             f"(world_size {world})" if red.active else "off (one rank)"),
                          exchange_inside_graph=bool(graphed is not None and graphed.exchange_captured),
                          volumes_per_s=round(world * batch / dt_step, 3), losses=[round(v, 4) for v in losses], phases=phases,
                          unet_fwd_bwd_breakdown={k: dict(launches=v["launches"], ms=round(v["ms"], 3), tflops=round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1))
                                                  for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:12]})))
if world > 1:
    dist.destroy_process_group()
