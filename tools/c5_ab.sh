cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py -k "bf16" tests/test_gpu_model.py -k "bf16" tests/test_gpu_configs.py -k "config4" -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -3
for v in 1 0; do
  VQVAE_Z16=$v python bench.py --workload c5 --bf16 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('c5 bf16 z16=$v: ms/step %.2f  losses %s gate %.1f us' % (d['ms_per_step'], d['losses_last_step'], 1e3*r['avg_launch_ms']))"
done
