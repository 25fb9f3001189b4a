#!/bin/bash
# Where the waves of the small-grid kernels spend their cycles: SQ counters (separate rocprofv3 --pmc passes, counters only) over the C3 latent UNet forward
# (tools/bench_c3_unet.py).  SQ_WAIT_ANY = wave parked (s_waitcnt / barrier), SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_ANY = issuing; the three add up to
# SQ_WAVE_CYCLES (quad-cycles).  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = share of LDS-array cycles lost to bank conflicts.  -> gpurun_out/pmc_small_grid/summary.txt
cd "$(dirname "$0")/.."
R=$PWD; OUT=$R/gpurun_out/pmc_small_grid; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
for C in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o c -- python $R/tools/bench_c3_unet.py > $OUT/$C.log 2>&1)
done
python - <<'PY' | tee $OUT/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_small_grid/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("C3 latent UNet forward (1x4x32^3 bf16), SQ counters per launch (mean over the launches of each kernel; quad-cycles unless noted)")
print(f"{'kernel':78s} {'launches':>8s} {'wave_cyc':>10s} {'parked':>7s} {'stall':>6s} {'issue':>6s} {'lds_conflict':>12s}")
rows = []
for k, d in agg.items():
    wc = d.get("SQ_WAVE_CYCLES")
    if not wc:
        continue
    m = lambda n: (sum(d[n]) / len(d[n])) if d.get(n) else float("nan")
    rows.append((sum(wc), k, len(wc), m("SQ_WAVE_CYCLES"), m("SQ_WAIT_ANY"), m("SQ_WAIT_INST_ANY"), m("SQ_ACTIVE_INST_ANY"), m("SQ_VALU_MFMA_BUSY_CYCLES"), m("SQ_BUSY_CYCLES"),
                 m("SQ_LDS_BANK_CONFLICT"), m("SQ_LDS_IDX_ACTIVE")))
for tot, k, n, wc, wa, wi, ai, mf, bz, bc, la in sorted(rows, reverse=True)[:14]:
    print(f"{k[:78]:78s} {n:8d} {wc:10.0f} {wa / wc:7.2f} {wi / wc:6.2f} {ai / wc:6.2f} {bc / la if la == la and la else float('nan'):12.3f}")
PY
