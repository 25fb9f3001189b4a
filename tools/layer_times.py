"""GPU: per-launch timing of one C2 UNet forward (HIP events around every launch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from generativemodels_amd import ops
from generativemodels_amd.networks.nets import DiffusionModelUNet
import bench
torch.manual_seed(0)
m = DiffusionModelUNet(**bench.C2).eval()
sd = bench.rerandomize_zero_params({k: v.clone() for k, v in m.state_dict().items()})
m.load_state_dict(sd)
m = m.to("cuda", torch.bfloat16)
x = torch.randn((1, 1, 128, 128, 128), device="cuda").bfloat16()
t = torch.tensor([500.0], device="cuda")
m(x, t); m(x, t); torch.cuda.synchronize()
ops.start_profile(); m(x, t); rec = ops.stop_profile()
tot = 0.0
for name, meta, ms in rec:
    tot += ms
    tf = meta["flops"] / max(ms, 1e-9) / 1e9
    gb = meta["bytes"] / max(ms, 1e-9) / 1e6
    print(f"{ms:8.3f} ms  {tf:8.1f} TF/s {gb:8.1f} GB/s  {name:32s} {meta.get('shape','')}")
print("sum of profiled launches", round(tot, 3), "ms over", len(rec), "launches")
