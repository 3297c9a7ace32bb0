cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q --no-header -p no:cacheprovider -k "pack_ahead or prepacked or graphed or streaming or one_sweep" 2>&1 | tail -8
for i in 1 2; do
for v in 0 1; do
  echo "PREPACK_CONVS=$v: $(VQVAE_PREPACK_CONVS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.3f ms  with_input %.3f (%s)" % (j["ms_per_step"], j["ms_per_step_with_input"], j["step_execution"][:10]))')"
  echo "PREPACK_CONVS=$v eager: $(VQVAE_PREPACK_CONVS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.3f ms  with_input %.3f (%s)" % (j["ms_per_step"], j["ms_per_step_with_input"], j["step_execution"][:10]))')"
done; done
