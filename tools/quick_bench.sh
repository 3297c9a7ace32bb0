#!/bin/bash
# bench.py once, the line's essentials + the per-kernel table.  usage: bash tools/quick_bench.sh [bench args]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/qb.json 2> gpurun_out/qb.err || tail -5 gpurun_out/qb.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/qb.json') if l.startswith('{')][-1])
r = d['roofline']
print('ms/step %.3f  with input %s  gate %.1f us frac %.3f' % (d['ms_per_step'], d.get('ms_per_step_with_input'), 1e3 * r['avg_launch_ms'], r['frac']))
for k in (r.get('kernels') or []):
    print('  %-46s n/step %5.1f avg %7.1f us %6.2f ms/step %s %.3f' % (k['name'][:46], k['launches_per_step'], 1e3 * k['avg_launch_ms'], k['ms_per_step'], k['bound'], k['frac']))
PY
