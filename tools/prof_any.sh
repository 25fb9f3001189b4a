#!/bin/bash
# rocprofv3 --kernel-trace --stats of any tool script: tools/prof_any.sh TAG SCRIPT [ARGS...]  ->  gpurun_out/TAG_kernel_stats.csv (+ the script's stdout in TAG_prof.log)
cd "$(dirname "$0")/.."
R=$PWD; TAG=$1; shift; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -rf $OUT/${TAG}_prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o p -- python $R/"$@" > $OUT/${TAG}_prof.log 2>&1)
F=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp $F $OUT/${TAG}_kernel_stats.csv && head -${PROF_HEAD:-25} $F | cut -c1-200
grep '^{' $OUT/${TAG}_prof.log | tail -2 | cut -c1-600
find $OUT/${TAG}_prof -name "*.csv" -size +3M -delete 2>/dev/null; find $OUT/${TAG}_prof -name "*.db" -delete 2>/dev/null
