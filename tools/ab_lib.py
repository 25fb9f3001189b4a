"""GPU: one line of JSON for whatever library GM_NATIVE_LIB selects -- C2 forward time (10 forwards between HIP events), the per-label kernel
sums of one profiled forward, and a digest of the output (variants of one kernel must agree bit for bit).
usage: GM_NATIVE_LIB=$PWD/generativemodels_amd/lib/libgmamd_eb.so python tools/ab_lib.py [tag]"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from generativemodels_amd import ops
from generativemodels_amd.networks.nets import DiffusionModelUNet

torch.manual_seed(0)
m = DiffusionModelUNet(**bench.C2).eval()
m.load_state_dict(bench.rerandomize_zero_params({k: v.clone() for k, v in m.state_dict().items()}))
m = m.to("cuda", torch.bfloat16)
x = torch.randn((1, 1, 128, 128, 128), generator=torch.Generator().manual_seed(7)).to("cuda", torch.bfloat16)
t = torch.tensor([500.0], device="cuda")
with torch.no_grad():
    y = m(x, t); m(x, t)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            m(x, t)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    ops.start_profile(); m(x, t); rec = ops.stop_profile()
agg = {}
for name, meta, ms in rec:
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += ms
print(json.dumps(dict(tag=sys.argv[1] if len(sys.argv) > 1 else "", lib=os.environ.get("GM_NATIVE_LIB", "default"), forward_ms=round(best, 3),
                      digest=hashlib.sha256(y.float().cpu().numpy().tobytes()).hexdigest()[:16],
                      kernels={k: [v[0], round(v[1], 3)] for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]})))
