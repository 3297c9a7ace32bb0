cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_model.py tests/test_gpu_configs.py -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tail -6
for i in 1 2; do
for v in 3 7; do
  echo "PRESPLIT=$v: $(VQVAE_PRESPLIT=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.3f ms  gate launch %.2f us  losses %s" % (j["ms_per_step"], 1e3*j["roofline"]["avg_launch_ms"], j["losses_last_step"]))')"
done; done
