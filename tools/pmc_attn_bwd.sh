#!/bin/bash
# HBM traffic of the fused attention backward at C2's size (one head x 32 768 tokens x 256 channels): FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 --pmc passes (counters only), against the composed path (score pass + weight-gradient kernels) in the same process.
cd "$(dirname "$0")/.."
R=$PWD; OUT=$R/gpurun_out/pmc_attn_bwd; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/attn_bwd_once.py <<'PY'
import math, sys, torch
sys.path.insert(0, sys.argv[1])
from generativemodels_amd import ops
l, dh = 32768, 256
g = torch.Generator().manual_seed(1)
q, k, v, go = (torch.randn((1, l, dh), generator=g).bfloat16().cuda() for _ in range(4))
lse = torch.empty((1, 1, l), dtype=torch.float32, device="cuda")
o = ops.attention(q, k, v, 1, 1 / math.sqrt(dh), lse_out=lse)
ops.attention_backward_fused(q, k, v, o, go, 1, 1 / math.sqrt(dh), lse=lse)
ops.attention_backward_bf16(q, k, v, o, go, 1, 1 / math.sqrt(dh))
torch.cuda.synchronize()
PY
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o c -- python /tmp/attn_bwd_once.py $R > $OUT/$C.log 2>&1)
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_attn_bwd/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel | launches | FETCH_SIZE KiB (sum over launches) | WRITE_SIZE KiB (sum) | HBM MB bracket [F + W, 2F + W] (FETCH_SIZE counts 64-B requests: a 128-B request counts half)")
tot = collections.defaultdict(lambda: [0.0, 0.0])
for k, d in sorted(agg.items(), key=lambda kv: -(sum(kv[1].get("FETCH_SIZE", [])) + sum(kv[1].get("WRITE_SIZE", [])))):
    f, w = sum(d.get("FETCH_SIZE", [])), sum(d.get("WRITE_SIZE", []))
    if f + w < 1024:
        continue
    print(f"{k[:90]:90s} | {max(len(d.get('FETCH_SIZE', [])), len(d.get('WRITE_SIZE', []))):3d} | {f:12.0f} | {w:12.0f} | [{(f + w) * 1024 / 1e6:9.1f}, {(2 * f + w) * 1024 / 1e6:9.1f}]")
    grp = "fused" if ("abd_" in k or "vt_pack_sets" in k) else ("composed" if ("attn_bwd_" in k or "wgrad" in k or "copy_channels" in k) else "other")
    tot[grp][0] += f; tot[grp][1] += w
for g, (f, w) in tot.items():
    print(f"== {g}: HBM MB bracket [{(f + w) * 1024 / 1e6:.1f}, {(2 * f + w) * 1024 / 1e6:.1f}]")
PY
