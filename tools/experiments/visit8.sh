cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 3000 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/gputest_full.txt 2>&1
tail -6 gpurun_out/gputest_full.txt
for v in "X=1" "VQVAE_FUSE_RELU_BWD=0"; do
env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('$v bench: ms/step %.3f | gate kernel avg %.1f us  frac %.3f  loss %s' % (d['ms_per_step'], 1e3 * r['avg_launch_ms'], r['frac'], d['losses_last_step']))"
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-graph > /tmp/b.log 2>&1
t=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
cd $GRAFT_REPO_ROOT
python tools/step_sequence.py $t > gpurun_out/step_sequence.txt 2>&1
head -1 gpurun_out/step_sequence.txt; grep -c ew_kernel gpurun_out/step_sequence.txt; grep wl1 gpurun_out/step_sequence.txt
