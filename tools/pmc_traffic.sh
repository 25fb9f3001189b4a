#!/bin/bash
# HBM traffic of the bench's kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (counters only).
cd "$(dirname "$0")/.."
R=$PWD; OUT=$R/gpurun_out/pmc_traffic; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o c -- python $R/bench.py --steps 1 --warmup 0 --graph 0 --cpu-baseline off --inference-steps 1 > $OUT/$C.log 2>&1)
done
python - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_traffic/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, d in agg.items():
    f, w = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
    if not f and not w:
        continue
    rows.append((k, len(f), sum(f) / max(len(f), 1), sum(w) / max(len(w), 1)))
rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
out = [dict(kernel=k, launches=n, fetch_kib_per_launch=round(f, 1), write_kib_per_launch=round(w, 1),
            hbm_mb_per_launch_corrected=round((2 * f + w) * 1024 / 1e6, 2)) for k, n, f, w in rows]
import subprocess, sys
sys.path.insert(0, ".")
from bench import kernel_source_sha
import os
head = os.environ.get("GM_GIT_HEAD") or subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None  # no .git on the GPU box
doc = dict(source_sha=kernel_source_sha(), git_head=head, command="bench.py --steps 1 --warmup 0 --graph 0 --cpu-baseline off --inference-steps 1",
           note="FETCH_SIZE doubled (gfx950: 128-B requests tallied at 64 B), WRITE_SIZE 1:1; separate --pmc passes", rows=out)
json.dump(doc, open("gpurun_out/pmc_traffic/summary.json", "w"), indent=1)
for o in out[:12]:
    print(o)
PY
