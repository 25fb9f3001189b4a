#!/bin/bash
# bench-only: does skewing the odd work-group slot of a CU against the even one change the tap-loop group times?  (timeline build)
cd "$(dirname "$0")/.."
for K in 0 6 12 24 48; do
  F=$((2048 + K * 65536)); [ $K == 0 ] && F=0
  echo "== skew $K x 64 cycles (flags $F)"
  GM_TL_FLAGS=$F GM_TL_SHAPES="192,64,128,11;64,64,128,11" GM_NATIVE_LIB=$PWD/generativemodels_amd/lib/libgmamd_timeline.so timeout 200 python tools/conv_timeline.py 2>/dev/null | grep -E "^---|chunk [0-4]: tap groups|tile top -> tile end"
done
