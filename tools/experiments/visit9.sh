cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_kernels.py -m gpu -q -x -s --no-header -p no:cacheprovider -k "dynamic_range or quiet" 2>&1 | grep -E "quiet|passed|failed|Error|assert" | head -20
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-graph > /tmp/b.log 2>&1
t=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
cd $GRAFT_REPO_ROOT
python tools/step_sequence.py $t > gpurun_out/step_sequence.txt 2>&1
head -1 gpurun_out/step_sequence.txt; grep wl1 gpurun_out/step_sequence.txt
