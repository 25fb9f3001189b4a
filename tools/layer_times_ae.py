"""GPU: per-launch times of AutoencoderKL.encode and .decode (the brain-bundle configuration of C3 / C4) at 1x1x256^3 in bf16 -- 66 % of a C4
training step (encode) and 23 % of a C3 volume (decode).   usage: python tools/layer_times_ae.py [edge]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import restatement as R
from generativemodels_amd import ops
from generativemodels_amd.networks.nets import AutoencoderKL
dev, dt = "cuda", torch.bfloat16
edge = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = dict(spatial_dims=3, in_channels=1, out_channels=1, latent_channels=4, num_channels=(64, 128, 128, 128), num_res_blocks=2,
           attention_levels=(False, False, False, False), with_encoder_nonlocal_attn=False, with_decoder_nonlocal_attn=False)
ae = AutoencoderKL(**cfg).eval()
ae.load_state_dict(R.synthetic_state_dict({k: tuple(v.shape) for k, v in ae.state_dict().items()}, seed=11))
ae = ae.to(dev, dt)
x = (torch.randn((1, 1, edge, edge, edge), generator=torch.Generator().manual_seed(23)) * 0.5).to(dev, dt)
torch.set_grad_enabled(False)
z, _ = ae.encode(x)
for name, fn in (("encode", lambda: ae.encode(x)), ("decode", lambda: ae.decode(z))):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"=== {name} 1x1x{edge}^3: {e0.elapsed_time(e1) / 5:.3f} ms (eager, 5 runs)")
    ops.start_profile(); fn(); rec = ops.stop_profile()
    flops = sum(m["flops"] for _, m, _ in rec)
    tot = sum(ms for _, _, ms in rec)
    for n, meta, ms in rec:
        print(f"  {ms:7.3f} ms {meta['flops'] / max(ms, 1e-9) / 1e9:8.1f} TF/s {meta['bytes'] / max(ms, 1e-9) / 1e6:8.1f} GB/s  {n:34s} {meta.get('shape', '')}")
    agg = collections.OrderedDict()
    for n, meta, ms in rec:
        a = agg.setdefault(n, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += meta["flops"]
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  sum {v[1]:8.3f} ms  x{v[0]:3d}  {v[2] / max(v[1], 1e-9) / 1e9:8.1f} TF/s  {k}")
    print(f"  {name}: sum of profiled launches {tot:.3f} ms over {len(rec)} launches, {flops / 1e12:.2f} TFLOP = {flops / tot / 1e9:.0f} TF/s")
