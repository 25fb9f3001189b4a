"""GPU: one forward of the C3 latent UNet (1x4x32^3, 41.7 M parameters, bf16) replayed from a HIP graph -- the 50x part of a latent-diffusion
sample, launch-latency-bound.  Prints ms per replayed forward.   usage: [GM_CONV_SK=0] [GM_CONV_SPLITK_WGS=..] python tools/bench_c3_unet.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import rerandomize_zero_params
from generativemodels_amd.networks.nets import DiffusionModelUNet
dev, dt = "cuda", torch.bfloat16
torch.manual_seed(0)
unet = DiffusionModelUNet(spatial_dims=3, in_channels=4, out_channels=4, num_channels=(64, 128, 256), attention_levels=(False, True, True),
                          num_res_blocks=2, num_head_channels=(0, 128, 256)).eval()
unet.load_state_dict(rerandomize_zero_params({k: v.clone() for k, v in unet.state_dict().items()}))
unet = unet.to(dev, dt)
x = torch.randn((1, 4, 32, 32, 32), generator=torch.Generator().manual_seed(7)).to(dev, dt)
t = torch.tensor([500.0], device=dev)
with torch.no_grad():
    for _ in range(3):
        ref = unet(x, t)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = unet(x, t)
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
print(json.dumps(dict(config="C3 latent UNet forward 1x4x32^3 bf16, HIP graph replay", ms_per_forward=round(e0.elapsed_time(e1) / reps, 4),
                      replay_equals_eager=bool(torch.equal(out, ref)), env={k: v for k, v in os.environ.items() if k.startswith("GM_CONV")})))
