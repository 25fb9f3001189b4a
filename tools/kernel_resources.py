"""CPU (build container): per-kernel register / spill / occupancy table of one HIP source, from hipcc's resource-usage remarks.
usage: python tools/kernel_resources.py conv_dma.hip [name filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "generativemodels_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = ["-ffp-contract=off"] if sys.argv[1] == "elementwise.hip" else []
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", *extra, "-x", "hip", "-c", src,
                    "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for c in rows:
    name = subprocess.run(["c++filt", c["name"]], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name:
        continue
    print(f"{name[:90]:90s} VGPR {c.get('VGPRs','?'):>4s} AGPR {c.get('AGPRs','?'):>3s} SGPR {c.get('TotalSGPRs','?'):>4s} spillV {c.get('VGPRs Spill','?'):>3s} spillS {c.get('SGPRs Spill','?'):>3s} "
          f"scratch {c.get('ScratchSize [bytes/lane]','?'):>4s} occ {c.get('Occupancy [waves/SIMD]','?')}")
if r.returncode != 0:
    print(r.stderr[-3000:])
