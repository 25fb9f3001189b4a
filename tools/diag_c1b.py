"""GPU: where a step of the C1b chain (2-D DDPM, 16 x 1 x 64 x 64, bf16, HIP-graph replay of the forward) goes: the replayed forward alone, the scheduler step alone
(CPU-generator noise + upload + one fused kernel), and both, per step.   usage: python tools/diag_c1b.py [bf16|fp32]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import rerandomize_zero_params
from generativemodels_amd.inferers import DiffusionInferer
from generativemodels_amd.networks.nets import DiffusionModelUNet
from generativemodels_amd.networks.schedulers import DDPMScheduler

dt = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "fp32") else torch.bfloat16
dev = "cuda"
torch.manual_seed(0)
torch.set_grad_enabled(False)
m = DiffusionModelUNet(2, 1, 1, num_channels=(32, 64), attention_levels=(False, True), num_res_blocks=1, num_head_channels=64).eval()
m.load_state_dict(rerandomize_zero_params({k: v.clone() for k, v in m.state_dict().items()}))
m = m.to(dev, dt)
sched = DDPMScheduler(1000)
sched.set_timesteps(1000)
x = torch.randn((16, 1, 64, 64), generator=torch.Generator().manual_seed(7)).to(dev, dt)
inf = DiffusionInferer(sched, use_hip_graph=True)
inf.sample(x, m, sched, verbose=False) if False else None
t = torch.full((16,), 500.0, device=dev)
# a captured forward, as the inferer builds it
g = torch.cuda.CUDAGraph()
for _ in range(3):
    y = m(x, t)
torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        y = m(x, t)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
N = 300


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3


print(f"{dt}: replayed forward alone        {timed(lambda: g.replay()):.4f} ms")
print(f"{dt}: eager forward alone           {timed(lambda: m(x, t)):.4f} ms")
print(f"{dt}: scheduler.step alone (t=500)  {timed(lambda: sched.step(y, 500, x)):.4f} ms")
print(f"{dt}: CPU randn of one step's noise {timed(lambda: torch.randn(tuple(x.shape), dtype=dt)):.4f} ms")
print(f"{dt}: ... + upload                  {timed(lambda: torch.randn(tuple(x.shape), dtype=dt).to(dev)):.4f} ms")
print(f"{dt}: replay + step                 {timed(lambda: (g.replay(), sched.step(y, 500, x))):.4f} ms")
