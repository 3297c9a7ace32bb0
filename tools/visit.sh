cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/kstats.sh --no-graph --no-fresh-input 2>&1 | head -22 | cut -c1-150
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'with input', d['ms_per_step_with_input'])"
