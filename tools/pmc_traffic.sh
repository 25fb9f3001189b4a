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
# Read-side calibration on this box (profiles/r03_fetch_size_calibration.json, tools/fetch_calib.hip): FETCH_SIZE = requests x 64 B; a request
# that IS 64 B -- the LDS-DMA convolutions' activation pieces: 64 B (32 bf16 channels) of a 128- or 256-byte voxel row -- is counted exactly
# (x1), a 128-B request of a wide streaming read is counted at half (x2, the guide's gfx950 note); WRITE_SIZE is exact.  A kernel mixing both
# (the convolutions also stream their weight panels, mostly L2 hits) lies between F + W and 2F + W; `corrected` takes the factor of the
# kernel's dominant read pattern and both ends of the bracket are kept.
def read_factor(kernel):
    return 1 if kernel.startswith("void conv_dma_kernel<") else 2
out = [dict(kernel=k, launches=n, fetch_kib_per_launch=round(f, 1), write_kib_per_launch=round(w, 1), read_factor=read_factor(k),
            hbm_mb_per_launch_corrected=round((read_factor(k) * f + w) * 1024 / 1e6, 2),
            hbm_mb_per_launch_bracket=[round((f + w) * 1024 / 1e6, 2), round((2 * f + w) * 1024 / 1e6, 2)]) for k, n, f, w in rows]
import subprocess, sys
sys.path.insert(0, ".")
from bench import kernel_source_sha
import os
head = os.environ.get("GM_GIT_HEAD") or subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None  # no .git on the GPU box
doc = dict(source_sha=kernel_source_sha(), git_head=head, command="bench.py --steps 1 --warmup 0 --graph 0 --cpu-baseline off --inference-steps 1",
           note="separate --pmc passes; FETCH_SIZE x read_factor (1 for the 64-byte activation pieces of the LDS-DMA convolutions, 2 for wide streaming reads: "
                "profiles/r03_fetch_size_calibration.json), WRITE_SIZE 1:1; bracket = [F + W, 2F + W]", rows=out)
json.dump(doc, open("gpurun_out/pmc_traffic/summary.json", "w"), indent=1)
for o in out[:12]:
    print(o)
PY
