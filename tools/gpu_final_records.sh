#!/bin/bash
# One GPU visit that produces the records of a finished tree: the PMC traffic passes that stamp profiles/pmc_hbm_traffic_current.json, the default bench (CPU leg on),
# rocprofv3 --kernel-trace --stats of the bench, per-launch layer times (C2, C3 latent UNet, 256^3 autoencoder), the other configurations' benches, the GPU suite.
#   usage: GM_GIT_HEAD=$(git rev-parse --short HEAD) tools/gpu_final_records.sh TAG [notests]
set -u
cd "$(dirname "$0")/.."
TAG=${1:-final}; OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=$OUT/${TAG}.log; : > $L
step() { echo "== $1 ($(date +%T))" >> $L; }
step pmc; timeout 900 bash tools/pmc_traffic.sh >> $L 2>&1; cp $OUT/pmc_traffic/summary.json $OUT/${TAG}_pmc_traffic.json 2>/dev/null; cp $OUT/pmc_traffic/summary.json profiles/pmc_hbm_traffic_current.json 2>/dev/null
# (the eager C4 step issues ~620 launches from Python and follows the host's clock: it runs BEFORE the 128-thread CPU-baseline leg of the bench, behind which it read 44-48 ms instead of 41)
step train; timeout 600 python tools/bench_train.py > $OUT/${TAG}_train.json 2> /dev/null; grep '^{' $OUT/${TAG}_train.json | head -c 900 >> $L; echo >> $L
step bench; timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?" >> $L; grep '^{' $OUT/${TAG}_bench.json | head -c 4500 >> $L; echo >> $L
step prof
rm -rf $OUT/${TAG}_prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/${TAG}_prof -o p -- python $OLDPWD/bench.py --steps 1 --warmup 1 --cpu-baseline off > $OLDPWD/$OUT/${TAG}_prof.log 2>&1)
F=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/${TAG}_kernel_stats.csv && head -14 $F | cut -c1-200 >> $L
find $OUT/${TAG}_prof -name "*.csv" -size +3M -delete 2>/dev/null; find $OUT/${TAG}_prof -name "*.db" -delete 2>/dev/null
step layers; timeout 300 python tools/layer_times.py > $OUT/${TAG}_layer_times_c2.txt 2>&1; tail -1 $OUT/${TAG}_layer_times_c2.txt >> $L
step ae256; timeout 300 python tools/layer_times_ae.py > $OUT/${TAG}_layer_times_ae.txt 2>&1; grep -i "encode\|decode\|sum " $OUT/${TAG}_layer_times_ae.txt | tail -8 >> $L
step c3; timeout 300 python tools/bench_c3.py > $OUT/${TAG}_c3.json 2> /dev/null; grep '^{' $OUT/${TAG}_c3.json | head -c 330 >> $L; echo >> $L
timeout 300 python tools/bench_c3_unet.py 2>/dev/null | tail -1 > $OUT/${TAG}_c3_unet.json; cat $OUT/${TAG}_c3_unet.json >> $L
step c1b; timeout 300 python tools/bench_c1b.py > $OUT/${TAG}_c1b.json 2> /dev/null; grep '^{' $OUT/${TAG}_c1b.json | head -c 600 >> $L; echo >> $L
step c5; timeout 300 python tools/bench_c5.py > $OUT/${TAG}_c5.json 2> /dev/null; grep '^{' $OUT/${TAG}_c5.json | head -c 330 >> $L; echo >> $L
if [ "${2:-}" != "notests" ]; then
  step full-tests; timeout 1900 python -m pytest tests -m gpu -q --maxfail=20 --durations=10 -p no:cacheprovider > $OUT/${TAG}_tests.log 2>&1; tail -16 $OUT/${TAG}_tests.log >> $L
  step smoke; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" >> $L 2>&1
fi
step done
tail -100 $L
