#!/bin/bash
# builds tools/attn_ablate_<mask>.bin for the masks given (default set) -- run them on the GPU box: for b in tools/attn_ablate_*.bin; do $b; done
cd "$(dirname "$0")/.."
for m in ${@:-0 1 8 16 24 25 32 2 6 7 31 63}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result -DGM_ATTN_ABLATE=$m tools/attn_ablate.hip generativemodels_amd/csrc/capi.cpp -o tools/attn_ablate_$m.bin 2>&1 | grep -E "error" -A3 | head -12 &
done
wait
ls -la tools/attn_ablate_*.bin | wc -l
