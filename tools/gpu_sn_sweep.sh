#!/bin/bash
# C3 latent UNet forward (graph replay) under the policy switches of tile configuration 24 (conv_sn.hip): off / contraction-depth bound / wave shape
cd "$(dirname "$0")/.."
run() { echo "$1: $(env $2 python tools/bench_c3_unet.py 2>/dev/null | tail -1 | cut -c1-130)"; }
run "SN off" GM_CONV_SN=0
for c in ${SN_CHUNKS:-4 6 8}; do run "8 waves, <= $c chunks" GM_CONV_SN_MAX_CHUNKS=$c; run "4 waves, <= $c chunks" "GM_CONV_SN_MAX_CHUNKS=$c GM_CONV_SN_WAVES=4"; done
run "SN off" GM_CONV_SN=0
