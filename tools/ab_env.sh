#!/bin/bash
# A/B of environment switches on one box: bash tools/ab_env.sh "<bench args>" "VAR=a VAR2=b" "VAR=c" ...   ("-" = no switch)
cd $GRAFT_REPO_ROOT
ARGS=$1; shift
for rep in 1 2; do
for e in "$@"; do
  if [ "$e" = "-" ]; then ee=""; else ee="$e"; fi
  env $ee python bench.py --no-cpu-baseline $ARGS | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('[$e]', round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms']*1e3,1), round(d['roofline']['frac'],4))"
done; done
