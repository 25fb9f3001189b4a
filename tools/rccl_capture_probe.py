"""Probe (GPU box, one rank): can a HIP graph capture RCCL all-reduces launched on a side stream from inside the captured region (the shape of
GradientReducer's in-backward bucket exchange)?  Prints what happened; exit code 0 either way."""
import os
import sys
import traceback

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
a = torch.ones(1 << 22, device=dev)
b = torch.ones(1 << 22, device=dev)
dist.all_reduce(a)  # communicator set-up outside the capture
torch.cuda.synchronize()
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        a.mul_(2.0)
        side.wait_event(torch.cuda.current_stream().record_event())
        with torch.cuda.stream(side):
            w = dist.all_reduce(a, async_op=True)
        b.add_(1.0)  # work of the capturing stream that may overlap the exchange
        w.wait()
        torch.cuda.current_stream().wait_stream(side)
        a.div_(1.0)
    print("capture: ok")
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("replay: ok, a[0] =", a[0].item(), "(expect 8.0), b[0] =", b[0].item(), "(expect 4.0)")
except Exception:
    print("capture / replay FAILED:")
    traceback.print_exc()
try:
    dist.destroy_process_group()
except Exception:
    pass
sys.exit(0)
