"""CPU (build container): where a kernel's scratch (spill) instructions sit relative to its MFMAs, per basic block.
usage: python tools/isa_spills.py conv_dma.hip <mangled-name substring>"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "generativemodels_amd", "csrc", sys.argv[1])
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-x", "hip", "-S", "--cuda-device-only", src,
                "-o", "/tmp/_isa.s"], capture_output=True, text=True)
s = open("/tmp/_isa.s").read()
for f in re.split(r"\n(?=_Z\w+:)", s):
    m = re.match(r"(_Z\w+):", f)
    if not m or sys.argv[2] not in m.group(1):
        continue
    lines = f.split("\n")
    cur, stats = "entry", collections.OrderedDict()
    stats[cur] = [0, 0, 0, 0, 0]
    for l in lines:
        if re.match(r"\.LBB\d+_\d+:", l):
            cur = l.split(":")[0]
            stats[cur] = [0, 0, 0, 0, 0]
        for i, pat in enumerate(("v_mfma", "scratch_", "s_barrier", "ds_read", "global_load_lds")):
            if pat in l:
                stats[cur][i] += 1
    print(m.group(1), len(lines), "lines")
    print("block                 mfma scratch barrier ds_read dma")
    for k, v in stats.items():
        if v[0] or v[1]:
            print(f"{k:20s} {v[0]:5d} {v[1]:7d} {v[2]:7d} {v[3]:7d} {v[4]:4d}")
