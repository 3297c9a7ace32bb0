# one eager kernel trace of the default bench (for tools/step_sequence.py)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/trace; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o tr -- python bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline > gpurun_out/trace/bench.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1); cp "$f" gpurun_out/trace/kernel_trace.csv; wc -l gpurun_out/trace/kernel_trace.csv; tail -1 gpurun_out/trace/bench.log | cut -c1-300
