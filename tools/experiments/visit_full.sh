cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q --no-header -p no:cacheprovider 2>&1 | tail -4
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for i in 1 2; do
  echo "bench: $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.3f ms  with_input %.3f gate launch %.2f us" % (j["ms_per_step"], j["ms_per_step_with_input"], 1e3*j["roofline"]["avg_launch_ms"]))')"
done
