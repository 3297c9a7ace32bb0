#!/bin/bash
# One GPU-box visit: the -m gpu suite, then the bench lines (c2 default, c4, c5 bf16); everything lands in gpurun_out/.
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider "$@" > gpurun_out/gputest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest.log
tail -40 gpurun_out/gputest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err; echo "bench c2 rc=$?"
timeout 300 python bench.py --workload c4 --steps 5 --warmup 2 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "bench c4 rc=$?"
timeout 300 python bench.py --workload c4 --vq-rows 1920 --steps 20 --warmup 3 > gpurun_out/bench_c4_small.json 2> gpurun_out/bench_c4_small.err; echo "bench c4 small rc=$?"
timeout 600 python bench.py --workload c5 --bf16 --steps 10 --warmup 3 > gpurun_out/bench_c5_bf16.json 2> gpurun_out/bench_c5_bf16.err; echo "bench c5 rc=$?"
for f in c2 c4 c4_small c5_bf16; do python - gpurun_out/bench_$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    r = d['roofline']
    print(sys.argv[1], 'ms/step %.3f value %.4g %s | roofline %.1f %s frac %.3f' % (d['ms_per_step'], d['value'], d['unit'], r['achieved'], r['unit'], r['frac']), d.get('cpu_baseline', {}).get('value'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
