"""GPU: per-launch times of one forward of the C1b 2-D UNet (16x1x64x64, attention at level 1; BASELINE configs[0]) -- the launch-latency-bound
2-D DDPM chain.   usage: python tools/layer_times_c1b.py"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import rerandomize_zero_params
from generativemodels_amd import ops
from generativemodels_amd.networks.nets import DiffusionModelUNet
dev, dt = "cuda", torch.bfloat16
torch.manual_seed(0)
unet = DiffusionModelUNet(2, 1, 1, num_channels=(32, 64), attention_levels=(False, True), num_res_blocks=1, num_head_channels=64).eval()
unet.load_state_dict(rerandomize_zero_params({k: v.clone() for k, v in unet.state_dict().items()}))
unet = unet.to(dev, dt)
x = torch.randn((16, 1, 64, 64), generator=torch.Generator().manual_seed(7)).to(dev, dt)
t = torch.full((16,), 500.0, device=dev)
torch.set_grad_enabled(False)
for _ in range(3):
    unet(x, t)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    unet(x, t)
e1.record(); torch.cuda.synchronize()
print("forward %.3f ms (eager, 10 runs)" % (e0.elapsed_time(e1) / 10))
ops.start_profile(); unet(x, t); rec = ops.stop_profile()
agg = collections.OrderedDict()
for name, meta, ms in rec:
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += ms
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1]:8.3f} ms  x{v[0]:3d}  avg {1e3 * v[1] / v[0]:7.1f} us  {k}")
print("sum of profiled launches %.3f ms over %d launches" % (tot, sum(v[0] for v in agg.values())))
if os.environ.get("GM_C3_PER_LAUNCH", "1") != "0":  # every launch in order, with its shape: which level the latency sits on
    print("-- per launch")
    for name, meta, ms in rec:
        tf = meta.get("flops", 0.0) / max(ms, 1e-9) / 1e9
        print(f"{1e3 * ms:8.1f} us {tf:8.1f} TF/s  {name:34s} {meta.get('shape', '')}")
