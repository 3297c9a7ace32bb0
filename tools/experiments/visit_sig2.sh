cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3; do
for v in 3 7; do
  echo "PRESPLIT=$v: $(VQVAE_PRESPLIT=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.3f ms  gate launch %.2f us" % (j["ms_per_step"], 1e3*j["roofline"]["avg_launch_ms"]))')"
done; done
