"""GPU, under rocprofv3 --kernel-trace: the C2 UNet forward either as eager launches or replayed from a HIP graph (`python tools/graph_gap_probe.py eager|graph`).
tools/graph_gaps.py reads the two kernel traces and compares the idle time BETWEEN consecutive kernels: why is the replay slower at this size?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import C2, rerandomize_zero_params
from generativemodels_amd.networks.nets import DiffusionModelUNet
from generativemodels_amd.inferers.inferer import _GraphedUNet
mode = sys.argv[1]
torch.set_grad_enabled(False)
torch.manual_seed(0)
m = DiffusionModelUNet(**C2).eval()
m.load_state_dict(rerandomize_zero_params({k: v.clone() for k, v in m.state_dict().items()}))
m = m.to("cuda", torch.bfloat16)
x = torch.randn((1, 1, 128, 128, 128), generator=torch.Generator().manual_seed(7)).to("cuda", torch.bfloat16)
t = torch.tensor([500.0], device="cuda")
for _ in range(2):
    m(x, t)
fn = (lambda: m(x, t)) if mode == "eager" else _GraphedUNet(m, x, t, None)
if mode == "graph":
    fn = (lambda g=fn: g(x, t))
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    fn()
e1.record(); torch.cuda.synchronize()
print(f"{mode}: {e0.elapsed_time(e1) / 5:.3f} ms per forward")
