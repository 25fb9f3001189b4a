"""CPU (build container): static instruction mix of one kernel per basic block (VALU / SALU / MFMA / LDS / VMEM / DMA / barriers / waits).
usage: python tools/isa_profile.py conv_dma.hip <mangled-name substring>"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "generativemodels_amd", "csrc", sys.argv[1])
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-x", "hip", "-S", "--cuda-device-only", src,
                "-o", "/tmp/_isa.s"], capture_output=True, text=True)
s = open("/tmp/_isa.s").read()
KEYS = ["valu", "salu", "mfma", "lds", "vmem", "dma", "barrier", "waitcnt", "branch"]
def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_load_lds"): return "dma"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "vmem"
    if op == "s_barrier": return "barrier"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    return None
for f in re.split(r"\n(?=_Z\w+:)", s):
    m = re.match(r"(_Z\w+):", f)
    if not m or sys.argv[2] not in m.group(1):
        continue
    cur, stats, loop = "entry", collections.OrderedDict(), {}
    stats[cur] = collections.Counter()
    for l in f.split("\n"):
        lm = re.match(r"(\.LBB\d+_\d+):(.*)", l)
        if lm:
            cur = lm.group(1)
            stats[cur] = collections.Counter()
            loop[cur] = "Loop" in lm.group(2)
            continue
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")):
            continue
        k = classify(t[0])
        if k:
            stats[cur][k] += 1
    tot = collections.Counter()
    print(m.group(1))
    print(f"{'block':14s} " + " ".join(f"{k:>7s}" for k in KEYS))
    for b, c in stats.items():
        tot.update(c)
        if sum(c.values()) >= 12:
            print(f"{b:14s} " + " ".join(f"{c[k]:7d}" for k in KEYS) + ("  (loop)" if loop.get(b) else ""))
    print(f"{'TOTAL':14s} " + " ".join(f"{tot[k]:7d}" for k in KEYS))
