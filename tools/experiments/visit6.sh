cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 3000 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/gputest_full.txt 2>&1
tail -6 gpurun_out/gputest_full.txt
for v in "X=1" "VQVAE_COND_KSTEP=0"; do
env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('$v bench: ms/step %.3f | gate kernel avg %.1f us  frac %.3f  loss %s' % (d['ms_per_step'], 1e3 * r['avg_launch_ms'], r['frac'], d['losses_last_step']))"
done
cp chainer-vq-vae_amd/libvqvae_hip.so /tmp/lib_keep.so
cp tools/experiments/abl/lib_phase.so chainer-vq-vae_amd/libvqvae_hip.so
python tools/experiments/phases.py 2>/dev/null | grep workgroups
cp /tmp/lib_keep.so chainer-vq-vae_amd/libvqvae_hip.so
bash tools/kstats.sh --no-graph 2>&1 | head -40 > gpurun_out/kstats_v6.txt; head -14 gpurun_out/kstats_v6.txt
