#!/bin/bash
# the wide token GEMM (small_ops.hip token_gemm_wide_kernel): kernel tests, then the C1b 2-D forward and the C3 latent UNet forward with the q | k | v projections
# on (a) the tiled kernel + V^T pack launch beyond 8192 rows, one block per wave below (the state before), (b) the wide form: blocks per wave by row count (the
# default), and from which row count on
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "token_gemm or transposed_v_image" 2>&1 | tail -3
c1b() { echo "C1b $1: $(env $2 python tools/diag_c1b.py 2>/dev/null | grep "replayed forward" | tr "\n" " ")"; }
c3() { echo "C3  $1: $(env $2 python tools/bench_c3_unet.py 2>/dev/null | tail -1 | cut -c60-110)"; }
for rep in 1 2; do
  c1b "before (<= 8192 rows, narrow)" "GM_TOKEN_GEMM_MAX_ROWS=8192 GM_TOKEN_GEMM_WIDE=0"
  c1b "wide, default" "GM_TOKEN_GEMM_WIDE=-1"
  c3 "before (narrow)" "GM_TOKEN_GEMM_WIDE=0"
  c3 "wide from 2048 rows (default)" "GM_TOKEN_GEMM_WIDE=-1"
  c3 "wide from 256 rows" "GM_TOKEN_GEMM_WIDE=256"
done
