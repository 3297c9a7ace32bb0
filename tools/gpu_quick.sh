#!/bin/bash
# quick GPU visit: kernel + model parity tests, bench line, single-stream kernel stats
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -6
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err; python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/bench_q.json') if l.startswith('{')][-1]); r = d['roofline']
print('bench: ms/step %.3f  %.4g samples/s | gate kernel %.1f TF frac %.3f avg %.1f us' % (d['ms_per_step'], d['value'], r['achieved'], r['frac'], 1e3 * r['avg_launch_ms']))
PY
bash tools/kstats.sh --no-overlap 2>&1 | head -16
