#!/bin/bash
# AddressSanitizer pass over the GPU kernel tests: tools/gpu_asan.sh TAG [pytest -k expression] [test files]
# Needs lib/libgmamd_asan.so (python -m generativemodels_amd._build --variant asan).  Kernels are built for gfx950:xnack+ and instrumented; every
# tensor is its own hipMalloc (PYTORCH_NO_CUDA_MEMORY_CACHING=1) so that the allocator's red zones sit right behind each operand.  This image's
# HIP runtime has no handler for the sanitizer's report hostcall: a violation ABORTS the process with "Hostcall: no handler found for service ID 4"
# (tools/asan/oob_probe.hip shows both outcomes); a clean run ends with pytest's summary.  The LDS-DMA requests are inline assembly: not instrumented.
set -u
cd "$(dirname "$0")/.."
TAG=${1:-asan}; K=${2:-}; shift; shift
FILES=${@:-tests/test_gpu_kernels.py}
CL=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)
export LD_PRELOAD=$CL/libclang_rt.asan-x86_64.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1
export HSA_XNACK=1 PYTORCH_NO_CUDA_MEMORY_CACHING=1
export GM_NATIVE_LIB=$PWD/generativemodels_amd/lib/libgmamd_asan.so
mkdir -p gpurun_out
( cd tools/asan && for n in 0 8; do timeout 60 ./oob_probe.bin $n 2>&1 | grep -c "no handler found for service ID 4" | sed "s/^/probe n=$n violations reported: /"; done ) > gpurun_out/${TAG}_asan.log 2>&1
if [ -n "$K" ]; then timeout 2400 python -m pytest $FILES -m gpu -q -x -p no:cacheprovider -k "$K" >> gpurun_out/${TAG}_asan.log 2>&1
else timeout 2400 python -m pytest $FILES -m gpu -q -x -p no:cacheprovider >> gpurun_out/${TAG}_asan.log 2>&1; fi
echo "pytest rc=$?" >> gpurun_out/${TAG}_asan.log
grep -c "no handler found for service ID 4" gpurun_out/${TAG}_asan.log | sed 's/^/violation messages in the log: /' >> gpurun_out/${TAG}_asan.log
tail -15 gpurun_out/${TAG}_asan.log
