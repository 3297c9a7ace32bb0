cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-graph > /tmp/b.log 2>&1
t=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
cd $GRAFT_REPO_ROOT
python tools/step_sequence.py $t > gpurun_out/step_sequence.txt 2>&1
python tools/phase_trace.py $t > gpurun_out/phase_trace.txt 2>&1
head -3 gpurun_out/step_sequence.txt; grep wl1 gpurun_out/step_sequence.txt
timeout 600 python -m pytest tests/test_gpu_bench_shapes.py -m gpu -q -x --no-header -p no:cacheprovider -k "presplit or resstack" 2>&1 | tail -3
