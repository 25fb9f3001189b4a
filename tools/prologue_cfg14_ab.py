"""GPU: the GroupNorm-apply + SiLU of a C2 ResnetBlock convolution -- (a) gm_gn_apply pass + the prologue-free cfg 14 launch (today's policy for large tensors) against
(b) cfg 14 with the transform applied IN LDS to the landed patch (one launch, no activated tensor in HBM) -- energy-metered like tools/taploop_energy.py
(ms per pair, mean W, J per pair).  Shapes: the C2 convolutions that sit behind a GroupNorm."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from generativemodels_amd import ops  # noqa: E402
from tools.taploop_energy import meter  # noqa: E402

dev = "cuda"
rows = []
for cin, cout, edge in ((64, 64, 128), (128, 64, 128), (128, 128, 64), (256, 128, 64)):
    x = torch.randn((1, edge, edge, edge, cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((cout, cin, 3, 3, 3), device=dev) / math.sqrt(cin * 27)).to(torch.bfloat16)
    b = torch.randn((cout,), device=dev)
    sc, sh = torch.rand((1, cin), device=dev) + 0.5, torch.randn((1, cin), device=dev) * 0.3
    act = torch.empty_like(x)
    flops = 2.0 * edge ** 3 * cin * cout * 27

    def two_pass():
        ops.gn_apply(x, sc, sh, "silu", out=act)
        ops.conv(act, w, b, kernel=3, padding=1, want_stats=True, force_cfg=14, ksplit=1)

    def fused():
        ops.conv(x, w, b, kernel=3, padding=1, want_stats=True, force_cfg=14, ksplit=1, pre=(sc, sh), pre_act="silu")

    def plain():
        ops.conv(act, w, b, kernel=3, padding=1, want_stats=True, force_cfg=14, ksplit=1)

    for name, fn in (("gn_apply pass + cfg 14", two_pass), ("cfg 14, transform in LDS", fused), ("cfg 14 alone (no GroupNorm)", plain)):
        rows.append(meter(f"{cin}->{cout} @{edge}^3 {name}", fn, flops, dict(cin=cin, cout=cout, edge=edge)))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "prologue_cfg14_ab.json"), "w"), indent=1)
