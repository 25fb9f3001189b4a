#!/bin/bash
# C3 (latent diffusion at 256^3): the fused in-LDS GroupNorm prologue policy A/B
cd "$(dirname "$0")/.."
for P in auto never always; do
  echo "== GM_DMA_FUSED_PROLOGUE=$P"
  GM_DMA_FUSED_PROLOGUE=$P timeout 300 python tools/bench_c3.py 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('sample_s','sample_s_hip_graph','decode_ms','latent_unet_forward_ms')})"
done
