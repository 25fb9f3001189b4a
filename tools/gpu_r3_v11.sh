#!/bin/bash
# round 3, GPU visit 11: per-kernel durations of the C3 latent UNet forward (1x4x32^3) -- rocprofv3 kernel trace + the tool's own event timing
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
timeout 300 python tools/layer_times_c3.py > $OUT/r3v11_layer_times_c3.txt 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/r3v11_prof -o c3 -- python $R/tools/layer_times_c3.py > $R/$OUT/r3v11_prof.log 2>&1)
F=$(find $OUT/r3v11_prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $OUT/r3v11_c3_kernel_stats.csv
rm -rf $OUT/r3v11_prof
