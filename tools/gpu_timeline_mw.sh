# cycle timelines of the LDS-DMA convolution tiles (cfg 14 / 21 / 22), plain, with operand traffic removed, and on all-zero operands (bench-only builds)
L=$PWD/generativemodels_amd/lib
S=${1:-"64,64,128,14;64,64,128,22;192,64,128,14;192,64,128,22"}
S2=${2:-"192,64,128,14;192,64,128,22"}
echo "=== timeline"; GM_TL_SHAPES="$S" GM_NATIVE_LIB=$L/libgmamd_timeline.so timeout 200 python tools/conv_timeline.py 2>&1 | grep -v amdgpu.ids
echo "=== timeline, zero operands"; GM_TL_ZERO=1 GM_TL_SHAPES="$S2" GM_NATIVE_LIB=$L/libgmamd_timeline.so timeout 200 python tools/conv_timeline.py 2>&1 | grep -v amdgpu.ids
for f in 512 1024 1536; do echo "=== timeline_ablate GM_TL_FLAGS=$f"; GM_TL_FLAGS=$f GM_TL_SHAPES="$S2" GM_NATIVE_LIB=$L/libgmamd_timeline_ablate.so timeout 200 python tools/conv_timeline.py 2>&1 | grep -v amdgpu.ids; done
echo "=== un-stamped, zero vs random operands (product library)"
BENCH_PLAIN=1 timeout 100 python tools/bench_conv.py 14,22 2>&1 | grep -v amdgpu.ids
BENCH_ZERO=1 BENCH_PLAIN=1 timeout 100 python tools/bench_conv.py 14,22 2>&1 | grep -v amdgpu.ids
