cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -k "vq" tests/test_gpu_configs.py -k "vq or config3 or codebook" -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -3
for v in 1 0; do
  VQVAE_VQ_CAND=$v python bench.py --workload c4 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('c4 cand=$v: ms/step %.2f  rows/s %.4g  rechecked %d  ok %s  TF %.1f' % (d['ms_per_step'], d['value'], d['rows_rechecked_exactly'], d['indices_match_reference_order_distance_on_64_rows'], r['achieved']))"
done
