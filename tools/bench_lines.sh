#!/bin/bash
# The round's bench lines, taken AFTER tools/restamp.sh stamped profiles/roofline_traffic.json on the final sources
# (so that `roofline.traffic` is non-null): default (configs[1]), c4, c5 --bf16 -> gpurun_out/lines_<tag>/
TAG=${1:-r5}
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/lines_$TAG; mkdir -p $OUT
python bench.py --steps 20 --warmup 5 2>$OUT/bench_default.err | tail -1 > $OUT/bench_default.json
python bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/c2_eager_bench_line.json
python bench.py --workload c4 --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/c4_bench_line.json
python bench.py --workload c5 --bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/c5_bf16_bench_line.json
python bench.py --workload c5 --steps 5 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/c5_bench_line.json
python - <<PY
import json
for n in ('bench_default', 'c2_eager_bench_line', 'c4_bench_line', 'c5_bf16_bench_line', 'c5_bench_line'):
    d = json.loads(open('$OUT/%s.json' % n).read()); r = d['roofline']
    print(n, 'ms/step %.2f' % d['ms_per_step'], 'with input', d.get('ms_per_step_with_input'), 'frac', r.get('frac'), 'traffic', r.get('traffic'), 'hbm', (r.get('hbm') or {}).get('frac'))
PY
