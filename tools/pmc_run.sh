#!/bin/bash
# rocprofv3 counter passes over a short bench run; prints per-kernel means for kernels matching $1.
# usage: bash tools/pmc_run.sh <kernel-substring> "<C1 C2 C3 C4>" ["<C5 ...>" ...]
cd /tmp && export TMPDIR=/tmp
pat="$1"; shift
i=0
for set in "$@"; do
  i=$((i+1)); rm -rf /tmp/pmc$i
  timeout 280 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pmc$i.log 2>&1 || echo "pass $i failed"
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc$i > /tmp/pmc$i.csv 2>/dev/null
  head -1 /tmp/pmc$i.csv; grep "$pat" /tmp/pmc$i.csv | head -6
done
