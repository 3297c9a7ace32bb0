cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do
for v in 8 1 0.25 0; do
  echo "MIN_GFLOP=$v: $(VQVAE_F16X2_MIN_GFLOP=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.3f ms  with_input %.3f (%s)" % (j["ms_per_step"], j["ms_per_step_with_input"], j["step_execution"][:10]))')"
done; done
