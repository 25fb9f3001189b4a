"""CPU (build container only -- needs /root/reference): wall time of the oracle restatement (oracle/restatement.py) against the UNMODIFIED
reference module on the same weights / input, config C2 at a reduced edge (default 64).  The ratio is what `cpu_baseline.kind = "port"` in
bench.py is worth relative to the reference's own code; it is recorded in DESIGN.md next to the baseline."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import restatement as R
from ref_loader import load_reference
from bench import C2, rerandomize_zero_params

size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
gen = load_reference()
torch.manual_seed(0)
ref = gen.networks.nets.DiffusionModelUNet(**C2).eval()
sd = rerandomize_zero_params({k: v.clone() for k, v in ref.state_dict().items()})
ref.load_state_dict(sd)
x = torch.randn((1, 1, size, size, size), generator=torch.Generator().manual_seed(7))
t = torch.tensor([500.0])
res = {}
with torch.no_grad():
    for name, fn in (("reference", lambda: ref(x, t)), ("restatement", lambda: R.unet_forward(sd, C2, x, t))):
        fn()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); y = fn(); ts.append(time.perf_counter() - t0)
        res[name] = sorted(ts)[1]
        res[name + "_out"] = y
err = (res.pop("reference_out") - res.pop("restatement_out")).abs().max().item()
print(json.dumps(dict(config=f"C2 forward, 1x1x{size}^3, fp32, {torch.get_num_threads()} threads", reference_s=round(res["reference"], 3),
                      restatement_s=round(res["restatement"], 3), ratio=round(res["restatement"] / res["reference"], 3), max_abs_diff=err)))
