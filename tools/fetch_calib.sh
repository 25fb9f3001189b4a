#!/bin/bash
# GPU: FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/fetch_calib.hip); writes gpurun_out/fetch_calib.json
cd "$(dirname "$0")/.."
R=$PWD; OUT=$R/gpurun_out/fetch_calib; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
BIN=$R/tools/fetch_calib.bin
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/fetch_calib.hip -o $BIN || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o c -- $BIN > $OUT/$C.log 2>&1)
done
python - <<'PY'
import csv, glob, collections, json
GIB2 = 2 << 30
known = {"calib_stream_read": ("read", GIB2), "calib_patch64_read<128>": ("read", GIB2 // 2), "calib_patch64_read<256>": ("read", GIB2 // 4),
         "calib_row128_read": ("read", GIB2), "calib_row128_write": ("write", GIB2)}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/fetch_calib/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, d in agg.items():
    key = next((n for n in known if n.split("<")[0] in k and (("<" not in n) or n.split("<")[1].rstrip(">") in k)), None)
    if key is None:
        continue
    kind, nbytes = known[key]
    f, w = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
    fk, wk = (sum(f) / len(f) if f else None), (sum(w) / len(w) if w else None)
    rows.append(dict(kernel=key, known_bytes=nbytes, kind=kind, fetch_size_kib=fk, write_size_kib=wk,
                     fetch_bytes_over_known=None if fk is None else round(fk * 1024 / nbytes, 4),
                     write_bytes_over_known=None if wk is None else round(wk * 1024 / nbytes, 4)))
doc = dict(note="rocprofv3 FETCH_SIZE / WRITE_SIZE (KiB) against known byte counts, 2 GiB buffer touched once per launch (8x the Infinity Cache); "
                "ratio = counter bytes / known bytes: a read pattern's HBM bytes = FETCH_SIZE / ratio", rows=rows)
json.dump(doc, open("gpurun_out/fetch_calib.json", "w"), indent=1)
for r in rows:
    print(r)
PY
