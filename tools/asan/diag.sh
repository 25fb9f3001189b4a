# AddressSanitizer + PyTorch diagnostics (round 4): shows why the pass runs through tools/asan/run_kernels.py -- under LD_PRELOAD of the sanitizer runtime
# the HIP runtime bundled with the PyTorch wheel fails its first device allocation ("AddressSanitizer: out of memory" at import torch)
CL=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)
export LD_PRELOAD=$CL/libclang_rt.asan-x86_64.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1
export HSA_XNACK=1 PYTORCH_NO_CUDA_MEMORY_CACHING=1
export GM_NATIVE_LIB=$PWD/generativemodels_amd/lib/libgmamd_asan.so
echo "--- python hello"; python -c "print('hello')"; echo "rc=$?"
echo "--- import torch"; timeout 200 python -c "import torch; print(torch.__version__, torch.cuda.is_available())"; echo "rc=$?"
echo "--- tensor on gpu"; timeout 200 python -c "import torch; x=torch.ones(4,device='cuda'); print((x+1).sum().item())"; echo "rc=$?"
echo "--- load lib + one op"; timeout 200 python -c "
import torch
from generativemodels_amd import ops
x=torch.randn(2,8,8,8,32,device='cuda').bfloat16()
sc=torch.ones(2,32,device='cuda'); sh=torch.zeros(2,32,device='cuda')
y=ops.gn_apply(x, sc, sh, 'silu'); torch.cuda.synchronize(); print('gn_apply ok', float(y.float().abs().mean()))
"; echo "rc=$?"
echo "--- pytest collect"; timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "gn_apply or layernorm" 2>&1 | tail -5; echo "rc=$?"
