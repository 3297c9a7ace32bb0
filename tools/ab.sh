#!/bin/bash
# A/B of env toggles in one box visit: tools/ab.sh "VAR=0" "VAR2=0" ...  (each run 3x, interleaved with the default)
cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f' % d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "default  $(run X=1)"
  for v in "$@"; do echo "$v  $(run $v)"; done
done
