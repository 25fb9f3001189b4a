"""GPU: capture forward_train + backward (+ Adam) of the C4 latent UNet in ONE HIP graph and replay it: gradients bitwise against the eager step,
time per replayed step against the eager step (host-bound: 621 launches at ~55 us of Python each).   usage: python tools/try_graph_train.py [small]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import generativemodels_amd as gm
from bench import rerandomize_zero_params
from generativemodels_amd.networks.nets import DiffusionModelUNet
small = len(sys.argv) > 1 and sys.argv[1] == "small"
dev = "cuda"
torch.manual_seed(0)
cfg = dict(spatial_dims=3, in_channels=4, out_channels=4, num_channels=(32, 64), attention_levels=(False, True), num_res_blocks=1,
           num_head_channels=(0, 32)) if small else \
      dict(spatial_dims=3, in_channels=4, out_channels=4, num_channels=(64, 128, 256), attention_levels=(False, True, True), num_res_blocks=2,
           num_head_channels=(0, 128, 256))
unet = DiffusionModelUNet(**cfg)
unet.load_state_dict(rerandomize_zero_params({k: v.clone() for k, v in unet.state_dict().items()}))
unet = unet.to(dev)
side = 16 if small else 32
x = torch.randn((1, 4, side, side, side), device=dev)
noise = torch.randn_like(x)
t = torch.tensor([500], device=dev)
params = [p for p in unet.parameters()]
def fwd_bwd():
    with gm.autocast(torch.bfloat16):
        pred = unet.forward_train(x, t)
    loss = F.mse_loss(pred.float(), noise)
    loss.backward()
    return loss
# eager reference gradients
for _ in range(2):
    unet.zero_grad(set_to_none=True); fwd_bwd()
ref = [None if p.grad is None else p.grad.clone() for p in params]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    unet.zero_grad(set_to_none=True); fwd_bwd()
torch.cuda.synchronize()
eager_ms = (time.perf_counter() - t0) / 5 * 1e3
print(f"eager forward_train + backward: {eager_ms:.2f} ms", flush=True)
# capture (PyTorch's whole-network pattern: warm up on a side stream, grads None so backward allocates them from the graph's pool)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        unet.zero_grad(set_to_none=True); fwd_bwd()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
unet.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
print("capturing ...", flush=True)
with torch.cuda.graph(g):
    static_loss = fwd_bwd()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
bad = 0
for p, r in zip(params, ref):
    if (p.grad is None) != (r is None):
        bad += 1
    elif r is not None and not torch.equal(p.grad, r):
        bad += 1
print(f"gradients after one replay: {len(params) - bad} of {len(params)} tensors bitwise equal to the eager step; loss {float(static_loss):.6f}", flush=True)
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print(f"replayed forward_train + backward: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per step (eager {eager_ms:.2f})", flush=True)
