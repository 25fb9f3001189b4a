#!/bin/bash
# rocprofv3 PMC passes over one convolution launch (counters only, no tracing domains). usage: pmc_conv.sh cin cout size cfg [flags] [prologue]
cd "$(dirname "$0")/.."
R=$PWD; OUT=$R/gpurun_out/pmc_conv; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
P3="SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -o c -- python $R/tools/one_conv.py "$@" > $OUT/p$i.log 2>&1)
done
python - <<'PY'
import csv, glob, collections
tot = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/pmc_conv/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if not any(k in r.get("Kernel_Name", "") for k in ("conv_fast", "conv_igemm", "conv_dma")):
            continue
        tot.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in tot.items():
    print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
