"""Backward-kernel micro-benchmark on MI355X (tools; not part of bench.py): weight gradient, data gradient (transposed convolution through
the forward kernels) and GroupNorm+SiLU backward at the C2 UNet's level shapes.  Prints one JSON line per measurement."""
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from generativemodels_amd import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return min(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))


def main():
    dtype = torch.bfloat16
    shapes = [(64, 64, 128, 1), (128, 64, 128, 1), (128, 128, 64, 1), (256, 256, 32, 1), (64, 128, 128, 2), (64, 64, 128, 2), (128, 128, 64, 2), (32, 32, 32, 1)]
    if len(sys.argv) > 1:
        shapes = shapes[: int(sys.argv[1])]
    out = []
    for cin, cout, size, stride in shapes:
        x = torch.randn((1, size, size, size, cin), device=DEV).to(dtype)
        so = size // stride
        gy = torch.randn((1, so, so, so, cout), device=DEV).to(dtype)
        w = (torch.randn((cout, cin, 3, 3, 3), device=DEV) * 0.05).to(dtype)
        flops = 2.0 * so ** 3 * cin * cout * 27
        t_w = timeit(lambda: ops.conv_wgrad(x, gy, 3, stride, 1))
        opad = size - ((so - 1) * stride - 2 + 3)
        t_d = timeit(lambda: ops.conv(gy, w, None, kernel=3, stride=stride, padding=1, transposed=True, output_padding=opad))
        t_f = timeit(lambda: ops.conv(x, w, None, kernel=3, stride=stride, padding=1))
        t_d2 = None
        if stride == 2:  # the path autograd takes: one sub-pixel launch on gy (ops.conv_stride2_dgrad); t_d above is the transposed-convolution path
            t_d2 = timeit(lambda: ops.conv_stride2_dgrad(gy, w, (size, size, size), 1))
        rec = dict(op="conv3x3x3", cin=cin, cout=cout, size=size, stride=stride, gflop=round(flops / 1e9, 1),
                   fwd_ms=round(t_f, 3), fwd_tflops=round(flops / t_f / 1e9, 1),
                   dgrad_ms=round(t_d, 3), dgrad_tflops=round(flops / t_d / 1e9, 1),
                   wgrad_ms=round(t_w, 3), wgrad_tflops=round(flops / t_w / 1e9, 1))
        if t_d2 is not None:
            rec["dgrad_subpixel_ms"] = round(t_d2, 3)
            rec["dgrad_subpixel_tflops"] = round(flops / t_d2 / 1e9, 1)
        if stride == 1:
            scale, shift = ops.gn_scale_shift_composed(x, 32, 1e-6, None, None)
            gx = torch.randn_like(x)
            t_g = timeit(lambda: ops.gn_backward(x, gx, scale, shift, None, 32, 1e-6, "silu"))
            rec["gn_bwd_ms"] = round(t_g, 3)
            rec["gn_bwd_gbs"] = round(x.numel() * 2 * 5 / t_g / 1e6, 1)  # x, gy read twice (stats + apply), dx written
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del x, gy, w
    return out


if __name__ == "__main__":
    main()


def attention_rows():
    """Fused flash backward vs the forward kernel at the training (C4) and sampling (C2) attention shapes."""
    dtype = torch.bfloat16
    import math
    for b, l, heads, dh in [(1, 4096, 1, 128), (1, 512, 1, 256), (1, 32768, 1, 256)]:
        c = heads * dh
        q, k, v, go = (torch.randn((b, l, c), device=DEV).to(dtype) for _ in range(4))
        scale = 1 / math.sqrt(dh)
        o = ops.attention(q, k, v, heads, scale)
        t_f = timeit(lambda: ops.attention(q, k, v, heads, scale), reps=3, warm=1)
        t_b = timeit(lambda: ops.attention_backward(q, k, v, o, go, heads, scale), reps=3, warm=1)
        f_fwd, f_bwd = 4.0 * b * heads * l * l * dh, 14.0 * b * heads * l * l * dh  # 2 / 7 GEMM units (S is recomputed twice)
        print(json.dumps(dict(op="attention", B=b, L=l, heads=heads, dh=dh, fwd_ms=round(t_f, 3), fwd_tflops=round(f_fwd / t_f / 1e9, 1),
                              bwd_ms=round(t_b, 3), bwd_tflops=round(f_bwd / t_b / 1e9, 1))), flush=True)


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "attn":
    attention_rows()
