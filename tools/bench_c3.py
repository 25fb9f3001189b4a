"""GPU: configuration C3 of SURVEY.md 8(d) -- latent diffusion at 256^3: 50 DDIM steps of the 41.7 M-parameter latent UNet on a
1x4x32^3 latent, then AutoencoderKL (brain-bundle widths 64/128/128/128) decode to 1x1x256^3, bf16, random-init weights.
Prints time per stage and the per-kernel breakdown of the decode (the 44 TFLOP part).   usage: python tools/bench_c3.py [size=256]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from bench import rerandomize_zero_params
from generativemodels_amd import ops
from generativemodels_amd.inferers import LatentDiffusionInferer
from generativemodels_amd.networks.nets import AutoencoderKL, DiffusionModelUNet
from generativemodels_amd.networks.schedulers import DDIMScheduler

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev, dt = "cuda", torch.bfloat16
torch.manual_seed(0)
ae = AutoencoderKL(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=4, num_channels=(64, 128, 128, 128), num_res_blocks=2,
                   attention_levels=(False, False, False, False), with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False).eval().to(dev, dt)
unet = DiffusionModelUNet(spatial_dims=3, in_channels=4, out_channels=4, num_channels=(64, 128, 256), attention_levels=(False, True, True),
                          num_res_blocks=2, num_head_channels=(0, 128, 256)).eval()
unet.load_state_dict(rerandomize_zero_params({k: v.clone() for k, v in unet.state_dict().items()}))
unet = unet.to(dev, dt)
sched = DDIMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0205, clip_sample=False)
sched.set_timesteps(50)
inf = LatentDiffusionInferer(sched, scale_factor=1.0)
lat = size // 8
noise = torch.randn((1, 4, lat, lat, lat), generator=torch.Generator().manual_seed(7)).to(dev, dt)


def timed(fn, reps=1):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


t_all, img = timed(lambda: inf.sample(noise, ae, unet, sched, verbose=False))
inf_g = LatentDiffusionInferer(sched, scale_factor=1.0, use_hip_graph=True)
t_graph, img_g = timed(lambda: inf_g.sample(noise, ae, unet, sched, verbose=False))
z = torch.randn((1, 4, lat, lat, lat), generator=torch.Generator().manual_seed(8)).to(dev, dt)
t_dec, _ = timed(lambda: ae.decode(z), 2)
x = torch.randn((1, 1, size, size, size), generator=torch.Generator().manual_seed(9)).to(dev, dt)
t_enc, _ = timed(lambda: ae.encode(x), 2)
t_unet, _ = timed(lambda: unet(noise, torch.tensor([500.0], device=dev)), 5)
ops.start_profile()
ae.decode(z)
rec = ops.stop_profile()
agg = {}
for name, meta, ms in rec:
    a = agg.setdefault(name, dict(launches=0, ms=0.0, flops=0.0))
    a["launches"] += 1; a["ms"] += ms; a["flops"] += meta["flops"]
print(json.dumps(dict(config=f"C3 at {size}^3", dtype="bf16", sample_s=round(t_all, 4), volumes_per_s=round(1 / t_all, 4), sample_s_hip_graph=round(t_graph, 4),
                      graph_vs_eager_maxdiff=float((img.float() - img_g.float()).abs().max()), decode_ms=round(t_dec * 1e3, 2),
                      encode_ms=round(t_enc * 1e3, 2), latent_unet_forward_ms=round(t_unet * 1e3, 3), output_finite=bool(torch.isfinite(img.float()).all()),
                      decode_breakdown={k: dict(launches=v["launches"], ms=round(v["ms"], 3), tflops=round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1))
                                        for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])})))
