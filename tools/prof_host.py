"""GPU: where the HOST time of a latent-UNet training step (C4: 1 x 4 x 32^3, mixed precision) goes -- cProfile of forward_train + backward, and
the wall time of the step against the sum of its kernels' durations.   usage: python tools/prof_host.py [top=40]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import generativemodels_amd as gm
from bench import rerandomize_zero_params
from generativemodels_amd import ops
from generativemodels_amd.networks.nets import DiffusionModelUNet
top = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = "cuda"
torch.manual_seed(0)
unet = DiffusionModelUNet(spatial_dims=3, in_channels=4, out_channels=4, num_channels=(64, 128, 256), attention_levels=(False, True, True),
                          num_res_blocks=2, num_head_channels=(0, 128, 256))
unet.load_state_dict(rerandomize_zero_params({k: v.clone() for k, v in unet.state_dict().items()}))
unet = unet.to(dev)
x = torch.randn((1, 4, 32, 32, 32), device=dev)
noise = torch.randn_like(x)
t = torch.tensor([500], device=dev)
def step():
    unet.zero_grad(set_to_none=True)
    with gm.autocast(torch.bfloat16):
        pred = unet.forward_train(x, t)
    F.mse_loss(pred.float(), noise).backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
host = (time.perf_counter() - t0) / 5
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 5
print(f"forward_train + backward: host {host * 1e3:.2f} ms per step, wall {wall * 1e3:.2f} ms")
ops.start_profile(); step(); rec = ops.stop_profile()
print(f"profiled launches: {len(rec)}, sum of kernel durations {sum(r[2] for r in rec):.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(top)
print(s.getvalue()[:9000])
