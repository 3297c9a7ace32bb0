#!/bin/bash
# End-of-round GPU visit: full -m gpu suite, smoke(), the profile set, default bench lines.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -4
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/profile_round.sh ${1:-r2} > gpurun_out/profile_round.log 2>&1; tail -3 gpurun_out/profile_round.log
