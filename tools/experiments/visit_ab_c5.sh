# same-box A/B of library builds on configs[4] bf16: bash tools/experiments/visit_ab_c5.sh name1 name2 ...
cd $GRAFT_REPO_ROOT; cp chainer-vq-vae_amd/libvqvae_hip.so /tmp/lib_keep.so
for rep in 1 2; do for n in "$@"; do
  if [ "$n" = product ]; then cp /tmp/lib_keep.so chainer-vq-vae_amd/libvqvae_hip.so; else cp tools/experiments/abl/lib_$n.so chainer-vq-vae_amd/libvqvae_hip.so; fi
  python bench.py --workload c5 --bf16 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$n', 'ms/step %.3f' % d['ms_per_step'], 'gate %.1f us' % (1e3*r['avg_launch_ms']))
for k in (r.get('kernels') or [])[:6]: print('   %-46s n/step %5.1f avg %7.1f us' % (k['name'][:46], k['launches_per_step'], 1e3*k['avg_launch_ms']))"
done; done
cp /tmp/lib_keep.so chainer-vq-vae_amd/libvqvae_hip.so
