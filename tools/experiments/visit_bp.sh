cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tail -3
for i in 1 2; do for v in 0 1; do
  echo "c5 bf16 BATCH_PULLBACK=$v: $(VQVAE_BATCH_PULLBACK=$v python bench.py --workload c5 --bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.3f ms" % j["ms_per_step"])')"
  echo "c2 x3   BATCH_PULLBACK=$v: $(VQVAE_BATCH_PULLBACK=$v python bench.py --matmul float32x3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.3f ms" % j["ms_per_step"])')"
done; done
