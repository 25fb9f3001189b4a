"""GPU: configuration C5 of SURVEY.md 8(d): VQVAE (8x down, 256 codes x 32) + DecoderOnlyTransformer(257, 4096, 256, 12, 8), raster-scan
ordering, sampling the 16^3 = 4096 latent tokens of one 128^3 volume with the KV-cache decoder, bf16, random-init weights.
Prints tokens/s, time per token, the decode time, and -- for a short prefix -- the cost of the reference's recompute-everything loop
on the same kernels (quadratic in the prefix length).   usage: python tools/bench_c5.py [tokens=4096] [graph]   ("graph": one HIP-graph replay per token, VQVAETransformerInferer(use_hip_graph=True))"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from generativemodels_amd.inferers import VQVAETransformerInferer
from generativemodels_amd.networks.nets import VQVAE, DecoderOnlyTransformer
from generativemodels_amd.utils import Ordering

ntok = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
side = round(ntok ** (1 / 3))
assert side ** 3 == ntok
dev, dt = "cuda", torch.bfloat16
torch.manual_seed(0)
vq = VQVAE(spatial_dims=3, in_channels=1, out_channels=1, num_embeddings=256, embedding_dim=32).eval().to(dev, dt)
tr = DecoderOnlyTransformer(num_tokens=257, max_seq_len=ntok, attn_layers_dim=256, attn_layers_depth=12, attn_layers_heads=8).eval().to(dev, dt)
order = Ordering("raster_scan", 3, (1, side, side, side))
use_graph = len(sys.argv) > 2 and sys.argv[2] == "graph"
inf = VQVAETransformerInferer(use_hip_graph=use_graph)
start = torch.full((1, 1), 256, device=dev)
torch.manual_seed(1)
inf.sample((2, 2, 2), start, vq, tr, Ordering("raster_scan", 3, (1, 2, 2, 2)), verbose=False)  # warm-up: packs weights
torch.cuda.synchronize()
t0 = time.perf_counter()
img = inf.sample((side, side, side), start, vq, tr, order, top_k=None, verbose=False)
torch.cuda.synchronize()
t_total = time.perf_counter() - t0
lat = torch.randint(0, 256, (1, side, side, side), device=dev)
vq.decode_samples(lat); torch.cuda.synchronize()
t0 = time.perf_counter(); vq.decode_samples(lat); torch.cuda.synchronize()
t_dec = time.perf_counter() - t0
# the reference loop (full forward of the growing prefix per token) on the same kernels, measured at a few prefix lengths
recompute = {}
for n in (256, 1024, min(4096, ntok)):
    x = torch.randint(0, 256, (1, n), device=dev)
    tr(x); torch.cuda.synchronize()
    t0 = time.perf_counter(); tr(x); torch.cuda.synchronize()
    recompute[n] = round((time.perf_counter() - t0) * 1e3, 3)
print(json.dumps(dict(config=f"C5: {ntok} tokens ({side}^3 latent of a {side * 8}^3 volume), transformer 12 x 256 x 8 heads", dtype="bf16", hip_graph=use_graph,
                      sample_s=round(t_total, 3), tokens_per_s=round(ntok / (t_total - t_dec), 1), ms_per_token=round((t_total - t_dec) * 1e3 / ntok, 4),
                      vqvae_decode_ms=round(t_dec * 1e3, 2), output_finite=bool(torch.isfinite(img.float()).all()),
                      full_forward_ms_at_prefix=recompute,
                      note="the reference's sampler runs one full forward of the whole prefix per token (sum over prefixes 1..N)")))
