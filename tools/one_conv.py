"""GPU: launch one convolution shape a few times (for rocprofv3 --pmc runs). usage: one_conv.py cin cout size cfg [flags] [prologue]"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from generativemodels_amd import ops
cin, cout, size, cfg = (int(a) for a in sys.argv[1:5])
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
pro = bool(int(sys.argv[6])) if len(sys.argv) > 6 else False
x = torch.randn((1, size, size, size, cin), device="cuda").bfloat16()
w = (torch.randn((cout, cin, 3, 3, 3), device="cuda") / math.sqrt(cin * 27)).bfloat16()
b = torch.randn((cout,), device="cuda")
pre = (torch.rand((1, cin), device="cuda") + 0.5, torch.randn((1, cin), device="cuda") * 0.1) if pro else None
ops._CONV_DEBUG_FLAGS = flags
for _ in range(3):
    ops.conv(x, w, b, kernel=3, padding=1, pre=pre, pre_act="silu" if pro else "none", force_cfg=cfg)
torch.cuda.synchronize()
