cd /root/repo
for G in 0 -1; do
  echo "== grid policy $G" 
  GM_CONV_DMA_GRID=$G timeout 300 python bench.py --steps 3 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_ddim_iteration'], d['unet_forward_ms'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done
echo "== timeline, policy 0"
GM_CONV_DMA_GRID=0 GM_TL_SHAPES="64,64,128,11;192,64,128,11" GM_NATIVE_LIB=$PWD/generativemodels_amd/lib/libgmamd_timeline.so timeout 200 python tools/conv_timeline.py 2>/dev/null
for F in 512 1024 1536; do
echo "== timeline ablate flags $F (512 no weights, 1024 no patch), policy 0"
GM_TL_FLAGS=$F GM_CONV_DMA_GRID=0 GM_TL_SHAPES="64,64,128,11;192,64,128,11" GM_NATIVE_LIB=$PWD/generativemodels_amd/lib/libgmamd_timeline_ablate.so timeout 200 python tools/conv_timeline.py 2>/dev/null
done
