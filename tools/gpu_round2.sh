#!/bin/bash
# One gpurun visit (round 2): stages t(ests) c(onv cfg check) p(mc of the conv kernel) b(ench) s(moke) k(ernel-trace profile) f(etch/write traffic)
# m = cycle timeline of the conv kernel (needs lib/libgmamd_timeline.so: python -m generativemodels_amd._build --variant timeline)
# usage: tools/gpu_round2.sh [stages] ; logs in gpurun_out/r2_*.log
set -u
cd "$(dirname "$0")/.."
STAGES="${1:-tcbs}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
LOG=$OUT/r2_round.log
echo "stages=$STAGES $(date)" > $LOG
rocminfo 2>/dev/null | grep -m2 -E "gfx|Compute Unit" >> $LOG
nproc >> $LOG
if [[ $STAGES == *s* ]]; then
  timeout 300 python __graft_entry__.py smoke > $OUT/r2_smoke.log 2>&1
  echo "smoke rc=$?" >> $LOG; tail -2 $OUT/r2_smoke.log >> $LOG
fi
if [[ $STAGES == *m* ]]; then
  GM_NATIVE_LIB=$PWD/generativemodels_amd/lib/libgmamd_timeline.so timeout 300 python tools/conv_timeline.py > $OUT/r2_timeline.log 2>&1
  echo "timeline rc=$?" >> $LOG; grep -A24 "64->64 @ 128^3 cfg11" $OUT/r2_timeline.log | grep -A14 "steady" >> $LOG
fi
if [[ $STAGES == *c* ]]; then
  timeout 900 python tools/check_conv_cfgs.py ${CONV_CFGS:-16,18,19} > $OUT/r2_conv_cfgs.log 2>&1
  echo "conv_cfgs rc=$?" >> $LOG; cat $OUT/r2_conv_cfgs.log >> $LOG
fi
if [[ $STAGES == *t* ]]; then
  timeout ${TEST_TIMEOUT:-2400} python -m pytest tests -m gpu -q --maxfail=40 --durations=15 -rP -p no:cacheprovider ${PYTEST_ARGS:-} > $OUT/r2_tests.log 2>&1
  echo "tests rc=$?" >> $LOG; grep -v "^\[parity\]\|^---\|^$\|Captured\|^_____\|PASSED" $OUT/r2_tests.log | tail -60 >> $LOG
  grep "\[parity\]" $OUT/r2_tests.log >> $LOG
fi
if [[ $STAGES == *p* ]]; then
  for CFG in ${PMC_CFGS:-11 18}; do
    P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
    P2="SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
    P3="SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"
    i=0
    for P in "$P1" "$P2" "$P3"; do
      i=$((i+1))
      D=$PWD/$OUT/r2_pmc_cfg${CFG}_p$i
      rm -rf $D
      (cd /tmp && timeout 300 rocprofv3 --pmc $P --output-format csv -d $D -o c -- python $OLDPWD/tools/one_conv.py ${PMC_SHAPE:-64 64 128} $CFG > $D.log 2>&1)
    done
    python - "$CFG" >> $LOG <<'PY'
import csv, glob, collections, sys
cfg = sys.argv[1]
tot = collections.OrderedDict()
for f in sorted(glob.glob(f"gpurun_out/r2_pmc_cfg{cfg}_p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "conv_dma" not in r.get("Kernel_Name", ""):
            continue
        tot.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print(f"--- PMC conv cfg{cfg}")
for k, v in tot.items():
    print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
  done
fi
if [[ $STAGES == *b* ]]; then
  timeout 1500 python bench.py ${BENCH_ARGS:---steps 3 --warmup 1} > $OUT/r2_bench.log 2>&1
  echo "bench rc=$?" >> $LOG; tail -c 6000 $OUT/r2_bench.log >> $LOG
fi
if [[ $STAGES == *a* ]]; then  # A/B: the same bench with the two-pass GroupNorm-apply (no in-LDS prologue)
  GM_DMA_FUSED_PROLOGUE=0 timeout 900 python bench.py --steps 2 --warmup 1 --cpu-baseline off > $OUT/r2_bench_twopass.log 2>&1
  echo "bench_twopass rc=$?" >> $LOG; tail -c 2500 $OUT/r2_bench_twopass.log >> $LOG
fi
if [[ $STAGES == *x* ]]; then  # A/B of conv debug flags (timing only)
  for FL in 0 2048; do
    CONV_FLAGS=$FL timeout 600 python tools/check_conv_cfgs.py ${CONV_CFGS:-16} --time-only > $OUT/r2_conv_flags_$FL.log 2>&1
    echo "conv flags $FL rc=$?" >> $LOG; grep -v amdgpu $OUT/r2_conv_flags_$FL.log >> $LOG
  done
fi
if [[ $STAGES == *n* ]]; then
  timeout 900 python tools/bench_attention.py > $OUT/r2_attention.log 2>&1
  echo "attention rc=$?" >> $LOG; cat $OUT/r2_attention.log >> $LOG
fi
if [[ $STAGES == *3* ]]; then
  timeout 900 python tools/bench_c3.py > $OUT/r2_c3.log 2>&1
  echo "c3 rc=$?" >> $LOG; tail -c 3000 $OUT/r2_c3.log >> $LOG
fi
if [[ $STAGES == *4* ]]; then  # C4: per-rank training step, plain and with the RCCL gradient exchange forced on one rank
  timeout 900 python tools/bench_train.py 256 1 bf16 3 > $OUT/r2_c4.log 2>&1
  echo "c4 rc=$?" >> $LOG; tail -c 3500 $OUT/r2_c4.log >> $LOG
  GM_FORCE_REDUCER=1 timeout 900 python tools/bench_train.py 256 1 bf16 3 > $OUT/r2_c4_rccl.log 2>&1
  echo "c4_rccl rc=$?" >> $LOG; tail -c 1200 $OUT/r2_c4_rccl.log >> $LOG
fi
if [[ $STAGES == *5* ]]; then
  timeout 900 python tools/bench_c5.py > $OUT/r2_c5.log 2>&1
  echo "c5 rc=$?" >> $LOG; tail -c 1500 $OUT/r2_c5.log >> $LOG
fi
if [[ $STAGES == *1* ]]; then
  timeout 900 python tools/bench_c1b.py > $OUT/r2_c1b.log 2>&1
  echo "c1b rc=$?" >> $LOG; tail -c 1500 $OUT/r2_c1b.log >> $LOG
fi
if [[ $STAGES == *l* ]]; then
  timeout 600 python tools/layer_times.py > $OUT/r2_layer_times.log 2>&1
  echo "layer_times rc=$?" >> $LOG; tail -1 $OUT/r2_layer_times.log >> $LOG
fi
if [[ $STAGES == *k* ]]; then
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/r2_prof -o bench -- python $OLDPWD/bench.py --steps 1 --warmup 0 --graph 0 --cpu-baseline off > $OLDPWD/$OUT/r2_prof.log 2>&1)
  echo "prof rc=$?" >> $LOG
  f=$(find $OUT/r2_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" >> $LOG
  find $OUT/r2_prof -name "*kernel_trace.csv" -size +20M -delete
fi
if [[ $STAGES == *f* ]]; then
  timeout 1500 bash tools/pmc_traffic.sh > $OUT/r2_pmc_traffic.log 2>&1
  echo "pmc_traffic rc=$?" >> $LOG; tail -15 $OUT/r2_pmc_traffic.log >> $LOG
fi
cat $LOG
