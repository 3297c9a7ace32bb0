# generic A/B: $1 = env var name toggled 0/1; tests named in $2 first
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
if [ -n "$2" ]; then timeout 1800 python -m pytest tests -m gpu -x -q --no-header -p no:cacheprovider -k "$2" 2>&1 | tail -5; fi
for i in 1 2; do
for v in 0 1; do
  echo "$1=$v: $(env $1=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.3f ms  with_input %.3f (%s)" % (j["ms_per_step"], j["ms_per_step_with_input"], j["step_execution"][:10]))')"
  echo "$1=$v eager: $(env $1=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.3f ms  with_input %.3f (%s)" % (j["ms_per_step"], j["ms_per_step_with_input"], j["step_execution"][:10]))')"
done; done
