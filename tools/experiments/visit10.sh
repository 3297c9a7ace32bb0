cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider -k "vq or VQ or config" 2>&1 | tail -6
for v in "X=1" "VQVAE_VQ_X2=0"; do
for rows in 1048560 1920; do
env $v timeout 300 python bench.py --workload c4 --vq-rows $rows --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v c4 rows $rows: ms/step %.3f' % d['ms_per_step'], d.get('roofline',{}).get('frac'))"
done; done
VQVAE_VQ_CAND_MINN=1024 timeout 300 python bench.py --workload c4 --vq-rows 1920 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('CAND_MINN=1024 c4 rows 1920: ms/step %.3f' % d['ms_per_step'])"
