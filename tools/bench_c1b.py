"""GPU: configuration C1b of SURVEY.md 8(d) (BASELINE.json configs[0]): 2-D DDPM, DiffusionModelUNet(2, 1, 1, (32, 64), attention at level 1,
1 res block, 64-wide heads) on N x 1 x 64 x 64 images, all 1000 DDPM steps, fp32 (the reference's precision for this config) and bf16,
eager launches vs HIP-graph replay of the forward.   usage: python tools/bench_c1b.py [batch=16] [steps=1000]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from bench import rerandomize_zero_params
from generativemodels_amd import host_noise
from generativemodels_amd.inferers import DiffusionInferer
from generativemodels_amd.networks.nets import DiffusionModelUNet
from generativemodels_amd.networks.schedulers import DDPMScheduler

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = "cuda"
torch.manual_seed(0)
base = DiffusionModelUNet(2, 1, 1, num_channels=(32, 64), attention_levels=(False, True), num_res_blocks=1, num_head_channels=64).eval()
base.load_state_dict(rerandomize_zero_params({k: v.clone() for k, v in base.state_dict().items()}))
out = dict(config=f"C1b: 2-D DDPM {steps} steps, {batch} x 1 x 64 x 64", results={})
for dt, name in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
    model = DiffusionModelUNet(2, 1, 1, num_channels=(32, 64), attention_levels=(False, True), num_res_blocks=1, num_head_channels=64).eval()
    model.load_state_dict(base.state_dict())
    model = model.to(dev, dt)
    sched = DDPMScheduler(1000)
    sched.set_timesteps(steps)
    noise = torch.randn((batch, 1, 64, 64), generator=torch.Generator().manual_seed(7)).to(dev, dt)
    host_noise.table_matches_torch()  # (the one-time check of the bf16 noise table against torch.randn: ~0.6 s, outside the timed chains)
    for graph, fast_noise, table in ((False, False, True), (True, False, True)) + (((True, False, False), (True, True, True)) if dt == torch.bfloat16 else ()):
        # bf16 rows: the default draws the reference's bf16 noise from the generator's bytes + a device table (host_noise.py); `_torch_randn_draw` is the
        # same chain through torch's serial bf16 fill (0.9 ms per step on this host; bit-identical images); `_fp32_noise_draw` is DDPMScheduler.fp32_noise_draw
        # (fp32 values rounded on the device: another stream)
        sched.fp32_noise_draw = fast_noise
        host_noise.ENABLED = table
        inf = DiffusionInferer(sched, use_hip_graph=graph)
        torch.manual_seed(1)
        inf.sample(noise, model, sched, verbose=False) if steps <= 50 else None
        torch.cuda.synchronize()
        torch.manual_seed(1)
        t0 = time.perf_counter()
        img = inf.sample(noise, model, sched, verbose=False)
        torch.cuda.synchronize()
        dtm = time.perf_counter() - t0
        out["results"][f"{name}{'_graph' if graph else ''}{'_fp32_noise_draw' if fast_noise else ''}{'' if table else '_torch_randn_draw'}"] = dict(seconds=round(dtm, 3), images_per_s=round(batch / dtm, 2),
                                                                 ms_per_step=round(dtm * 1e3 / steps, 4), finite=bool(torch.isfinite(img.float()).all()),
                                                                 checksum=float(img.float().abs().sum().item()))
print(json.dumps(out))
