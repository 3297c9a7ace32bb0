cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_bench_shapes.py -m gpu -q -x --no-header -p no:cacheprovider -k "presplit or resstack_b16 or resblock_b16" 2>&1 | tail -25
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err; tail -3 gpurun_out/bench_q.err; python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/bench_q.json') if l.startswith('{')][-1]); r = d['roofline']
print('bench: ms/step %.3f  %.4g samples/s | gate kernel %.1f TF frac %.3f avg %.1f us' % (d['ms_per_step'], d['value'], r['achieved'], r['frac'], 1e3 * r['avg_launch_ms']))
PY
VQVAE_PRESPLIT=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('PRESPLIT=0 bench: ms/step %.3f | gate kernel avg %.1f us' % (d['ms_per_step'], 1e3 * r['avg_launch_ms']))"
bash tools/kstats.sh --no-graph 2>&1 | head -14
