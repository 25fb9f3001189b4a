#!/bin/bash
# AddressSanitizer pass over the C-ABI kernels: tools/gpu_asan.sh TAG
# Needs lib/libgmamd_asan.so linked against the SYSTEM HIP runtime:
#   GM_HIP_RUNTIME=/opt/rocm/lib/libamdhip64.so python -m generativemodels_amd._build --variant asan
# (kernels for gfx950:xnack+, every translation unit without LDS-DMA inline assembly instrumented).  The driver is tools/asan/run_kernels.py:
# numpy + ctypes, no torch -- the sanitizer's hsa_amd_memory_pool_allocate interceptor fails every allocation of the HIP runtime bundled with
# the PyTorch wheel.  This image's HIP runtime has no handler for the sanitizer's report hostcall: a violation ABORTS the process with
# "Hostcall: no handler found for service ID 4" (tools/asan/oob_probe.hip shows both outcomes); a clean run ends with the driver's summary line.
set -u
cd "$(dirname "$0")/.."
TAG=${1:-asan}
CL=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)
export LD_PRELOAD=$CL/libclang_rt.asan-x86_64.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1
export HSA_XNACK=1
export GM_NATIVE_LIB=$PWD/generativemodels_amd/lib/libgmamd_asan.so
mkdir -p gpurun_out
LOG=gpurun_out/${TAG}_asan.log
( cd tools/asan && for n in 0 8; do timeout 60 ./oob_probe.bin $n 2>&1 | grep -c "no handler found for service ID 4" | sed "s/^/probe (oob_probe.bin $n): sanitizer aborts reported: /"; done ) > $LOG 2>&1
GM_ASAN_VERBOSE=${GM_ASAN_VERBOSE:-} timeout 1500 python tools/asan/run_kernels.py >> $LOG 2>&1
echo "driver rc=$?" >> $LOG
grep -c "no handler found for service ID 4" $LOG | sed 's/^/sanitizer abort messages in the log (the probe accounts for 1): /' >> $LOG
tail -25 $LOG
