#!/bin/bash
# usage: res.sh PART [extra flags]  -> per-kernel resource usage of conv_dma.hip part
P=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -x hip -S --cuda-device-only -DGM_DMA_PART=$P "$@" /root/repo/generativemodels_amd/csrc/conv_dma.hip -o /tmp/p$P.s -Rpass-analysis=kernel-resource-usage 2> /tmp/p$P.remarks
python - /tmp/p$P.remarks <<'PY'
import re,sys
txt=open(sys.argv[1]).read()
if "error:" in txt: print(txt[-3000:])
for line in txt.splitlines():
    m=re.search(r"remark: (?:[^:]*:\d+:\d+: +)?(.*?) \[-Rpass",line)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith("Function Name") or t.startswith("Name:"):
        print(); print(t.split("kernel")[1][:40],end=" | ")
    elif any(k in t for k in ("VGPRs","Spill","Scratch","SGPRs:")):
        print(t.replace("[bytes/lane]",""),end=" | ")
print()
PY
