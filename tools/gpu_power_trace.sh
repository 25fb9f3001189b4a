# GPU power / clock trace during the C2 bench: is the chip at its power cap? (rocm-smi sampled every 0.25 s next to bench.py)
OUT=gpurun_out/${1:-r04}_power_trace.txt
( rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^$" | head -30 ) > $OUT
echo "=== idle sample above; bench starts" >> $OUT
( ${2:-python bench.py --steps 6 --warmup 1 --cpu-baseline off} > gpurun_out/${1:-r04}_power_bench.json 2>/dev/null ) &
BP=$!
for i in $(seq 1 80); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' ' >> $OUT; echo >> $OUT
  sleep 0.25
  kill -0 $BP 2>/dev/null || break
done
wait $BP
echo "=== bench line" >> $OUT; tail -1 gpurun_out/${1:-r04}_power_bench.json | cut -c1-300 >> $OUT
