cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/trace; export TMPDIR=/tmp
for v in 0 1; do
rm -rf /tmp/tr
VQVAE_DEFER_WGRAD=$v rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o tr -- python bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline > gpurun_out/trace/bench_$v.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1); cp "$f" gpurun_out/trace/kernel_trace_defer$v.csv
done
