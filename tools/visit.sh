cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for f in "" "--no-graph"; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $f 2>gpurun_out/bench.err | tail -1 > gpurun_out/bench_q.json; tail -3 gpurun_out/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_q.json').read()); r = d['roofline']
print(d['step_execution'][:30], 'bench: ms/step %.3f (with input %s)  %.4g samples/s | gate kernel %.1f TF frac %.3f avg %.1f us' % (d['ms_per_step'], d.get('ms_per_step_with_input'), d['value'], r['achieved'], r['frac'], 1e3 * r['avg_launch_ms']))
PY
done
