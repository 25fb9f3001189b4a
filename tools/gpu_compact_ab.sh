cd /root/repo
for rep in 1 2; do
for th in 256 1024 2048 100000; do
  echo -n "above=$th  C3 latent: "; GM_STATS_COMPACT_ABOVE=$th timeout 120 python tools/bench_c3_unet.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_forward'], d['replay_equals_eager'])"
  echo -n "above=$th  C2 bench:  "; GM_STATS_COMPACT_ABOVE=$th timeout 200 python bench.py --steps 2 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_ddim_iteration'], d['unet_forward_ms'])"
done; done
