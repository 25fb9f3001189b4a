#!/bin/bash
# round 5, final GPU visit: smoke, the default bench (CPU leg on: kind = reference through oracle/_ref), rocprofv3 --kernel-trace --stats of the bench,
# the PMC traffic passes that stamp profiles/pmc_hbm_traffic_current.json
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=$OUT/r5final.log; : > $L
step() { echo "== $1 ($(date +%T))" >> $L; }
step smoke; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r5final_smoke.log 2>&1; tail -2 $OUT/r5final_smoke.log >> $L
step pmc; timeout 900 bash tools/pmc_traffic.sh >> $L 2>&1; cp $OUT/pmc_traffic/summary.json $OUT/r5final_pmc_traffic.json 2>/dev/null; cp $OUT/pmc_traffic/summary.json profiles/pmc_hbm_traffic_current.json 2>/dev/null
step bench; timeout 900 python bench.py > $OUT/r5final_bench.json 2> $OUT/r5final_bench.err; echo "bench rc=$?" >> $L; grep '^{' $OUT/r5final_bench.json | head -c 6000 >> $L; echo >> $L
step prof
rm -rf $OUT/r5final_prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/r5final_prof -o p -- python $OLDPWD/bench.py --steps 1 --warmup 1 --cpu-baseline off > $OLDPWD/$OUT/r5final_prof.log 2>&1)
F=$(find $OUT/r5final_prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/r5final_kernel_stats.csv && head -12 $F | cut -c1-200 >> $L
find $OUT/r5final_prof -name "*.csv" -size +3M -delete 2>/dev/null; find $OUT/r5final_prof -name "*.db" -delete 2>/dev/null
step done
tail -60 $L
