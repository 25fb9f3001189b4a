export GM_NATIVE_LIB=$PWD/generativemodels_amd/lib/libgmamd_ablate.so BENCH_PLAIN=1
for f in 0 512 1024 1536 256 1792; do echo "== CONV_FLAGS=$f"; CONV_FLAGS=$f timeout 120 python tools/bench_conv.py 14,21 2>&1 | grep -v amdgpu.ids; done
