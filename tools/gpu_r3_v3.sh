#!/bin/bash
# round 3, GPU visit 3: A/B of the early-barrier conv variants, diagnostics of the marching out-head kernel, the tests added since visit 2
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
LOG=$OUT/r3v3_round.log
echo "$(date)" > $LOG
L=$PWD/generativemodels_amd/lib
for V in main eb ebi main eb ebi; do
  if [ $V = main ]; then unset GM_NATIVE_LIB; else export GM_NATIVE_LIB=$L/libgmamd_$V.so; fi
  timeout 300 python tools/ab_lib.py $V >> $OUT/r3v3_ab.jsonl 2>> $OUT/r3v3_ab.err
done
unset GM_NATIVE_LIB
cat $OUT/r3v3_ab.jsonl >> $LOG
timeout 300 python tools/diag_cout1.py diff > $OUT/r3v3_cout1_diff.txt 2>&1
timeout 300 python tools/diag_cout1.py > $OUT/r3v3_cout1_time.txt 2>&1
for LTD in 2 3 4; do GM_CONV_COUT1_LTD=$LTD timeout 200 python tools/diag_cout1.py >> $OUT/r3v3_cout1_time.txt 2>&1; done
cat $OUT/r3v3_cout1_diff.txt $OUT/r3v3_cout1_time.txt >> $LOG
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_kernels.py -m gpu -q -k "spade or marching or subpixel or bitwise or vqvae" -p no:cacheprovider > $OUT/r3v3_tests.log 2>&1
echo "tests rc=$?" >> $LOG; tail -15 $OUT/r3v3_tests.log >> $LOG
(cd /tmp && timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d $PWD/../$OUT/r3v3_rdreq -o c -- $OLDPWD/tools/fetch_calib.bin > /dev/null 2>&1)
python - <<'PY' >> $LOG 2>&1
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r3v3_rdreq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k[:60], {c: sum(v) / len(v) for c, v in d.items()})
PY
echo "done $(date)" >> $LOG
