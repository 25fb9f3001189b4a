cd /tmp; export TMPDIR=/tmp
for m in eager graph; do rm -rf $GRAFT_REPO_ROOT/gpurun_out/r04v16_gap_$m; timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04v16_gap_$m -o t -- python $GRAFT_REPO_ROOT/tools/graph_gap_probe.py $m 2>&1 | grep "per forward"; done
cd $GRAFT_REPO_ROOT
python tools/graph_gaps.py $(find gpurun_out/r04v16_gap_eager -name "*kernel_trace.csv" | head -1) $(find gpurun_out/r04v16_gap_graph -name "*kernel_trace.csv" | head -1) | tee gpurun_out/r04v16_graph_gaps.txt
rm -rf gpurun_out/r04v16_gap_eager gpurun_out/r04v16_gap_graph
timeout 900 bash tools/gpu_asan.sh r04v16 "" tests/test_gpu_kernels.py
