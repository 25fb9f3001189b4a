#!/bin/bash
# rocprofv3 PMC pass: instruction-cache requests / hits / misses of one convolution launch.  usage: pmc_icache.sh cin cout size cfg
cd "$(dirname "$0")/.."
R=$PWD; OUT=$R/gpurun_out/pmc_icache; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for P in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -o c -- python $R/tools/one_conv.py "$@" > $OUT/p$i.log 2>&1)
  tail -2 $OUT/p$i.log | cut -c1-200
done
python - <<'PY'
import csv, glob, collections
tot = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/pmc_icache/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if not any(k in r.get("Kernel_Name", "") for k in ("conv_fast", "conv_igemm", "conv_dma")):
            continue
        tot.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in tot.items():
    print(f"{k:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
