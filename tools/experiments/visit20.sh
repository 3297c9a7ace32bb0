cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_dp.py tests/test_gpu_bench_shapes.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -4
for i in 1 2 3; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('bench: ms/step %.3f (with input %.3f) | gate kernel avg %.1f us  loss %s' % (d['ms_per_step'], d['ms_per_step_with_input'], 1e3 * r['avg_launch_ms'], d['losses_last_step']))"
done
bash tools/kstats.sh --no-graph 2>&1 | grep -E "sum of"; grep -E "pullback" gpurun_out/kstats.csv | cut -c1-120
