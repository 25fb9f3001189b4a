#!/bin/bash
# One GPU visit that produces the records of a finished tree (round 5's last visit, kept as run): a kernel-test subset, the PMC traffic passes that stamp
# profiles/pmc_hbm_traffic_current.json, the default bench (CPU leg on), rocprofv3 --kernel-trace --stats of the bench, the 256^3 autoencoder layer times, the GPU suite.
# (The round's A/B visits were one-off scripts; what each ran is in the header of the profiles/r05_* file it produced.  tools/gpu_visit.sh is the parameterised form.)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=$OUT/r5final2.log; : > $L
step() { echo "== $1 ($(date +%T))" >> $L; }
step kernel-tests; timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -p no:cacheprovider -k "cin or conv_in or edge or geometries or small_cin" > $OUT/r5final2_ktests.log 2>&1; tail -3 $OUT/r5final2_ktests.log >> $L
step pmc; timeout 900 bash tools/pmc_traffic.sh >> $L 2>&1; cp $OUT/pmc_traffic/summary.json $OUT/r5final2_pmc_traffic.json 2>/dev/null; cp $OUT/pmc_traffic/summary.json profiles/pmc_hbm_traffic_current.json 2>/dev/null
step bench; timeout 900 python bench.py > $OUT/r5final2_bench.json 2> $OUT/r5final2_bench.err; echo "bench rc=$?" >> $L; grep '^{' $OUT/r5final2_bench.json | head -c 3000 >> $L; echo >> $L
step prof
rm -rf $OUT/r5final2_prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/r5final2_prof -o p -- python $OLDPWD/bench.py --steps 1 --warmup 1 --cpu-baseline off > $OLDPWD/$OUT/r5final2_prof.log 2>&1)
F=$(find $OUT/r5final2_prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/r5final2_kernel_stats.csv && head -12 $F | cut -c1-200 >> $L
find $OUT/r5final2_prof -name "*.csv" -size +3M -delete 2>/dev/null; find $OUT/r5final2_prof -name "*.db" -delete 2>/dev/null
step ae256; timeout 300 python tools/layer_times_ae.py > $OUT/r5final2_layer_times_ae.txt 2>&1; grep -i "encode\|decode\|cfg12\|sum " $OUT/r5final2_layer_times_ae.txt | tail -20 >> $L
step full-tests; timeout 1700 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $OUT/r5final2_tests.log 2>&1; tail -4 $OUT/r5final2_tests.log >> $L
step done
tail -70 $L
