cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 3000 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/gputest_full.txt 2>&1
tail -4 gpurun_out/gputest_full.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_default.json; python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_default.json').read()); r = d['roofline']
print('bench: ms/step %.3f (with input %.3f) | gate %.1f us frac %.3f | cpu %s' % (d['ms_per_step'], d['ms_per_step_with_input'], 1e3 * r['avg_launch_ms'], r['frac'], d.get('cpu_baseline', {}).get('value')))
PY
