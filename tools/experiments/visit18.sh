cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider -k "bf16 or config4 or configs4 or c5" 2>&1 | tail -4
for v in "X=1" "VQVAE_FUSE_PULLBACK=0"; do
env $v timeout 300 python bench.py --workload c5 --bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('$v c5 bf16: ms/step %.3f | gate kernel avg %.1f us  loss %s' % (d['ms_per_step'], 1e3 * r['avg_launch_ms'], d['losses_last_step']))"
done
