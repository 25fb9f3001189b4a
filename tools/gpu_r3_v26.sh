#!/bin/bash
# round 3, GPU visit 26: the measurement records of the final kernels -- PMC traffic pass (FETCH_SIZE / WRITE_SIZE, separate passes), rocprofv3
# kernel statistics of the bench command, the default bench run (with the CPU baseline), smoke
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
LOG=$OUT/r3v26_round.log
echo "$(date)" > $LOG
timeout 900 bash tools/pmc_traffic.sh > $OUT/r3v26_pmc.log 2>&1; echo "pmc rc=$?" >> $LOG; tail -14 $OUT/r3v26_pmc.log >> $LOG
cp $OUT/pmc_traffic/summary.json $OUT/r3v26_pmc_traffic_summary.json 2>/dev/null
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/r3v26_prof -o bench -- python $R/bench.py --steps 1 --warmup 0 --graph 0 --cpu-baseline off > $R/$OUT/r3v26_prof.log 2>&1)
F=$(find $OUT/r3v26_prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $OUT/r3v26_bench_kernel_stats.csv && head -8 $F | cut -c1-160 >> $LOG
rm -rf $OUT/r3v26_prof
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r3v26_smoke.log 2>&1; tail -2 $OUT/r3v26_smoke.log >> $LOG
echo "done $(date)" >> $LOG
