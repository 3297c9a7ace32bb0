#!/bin/bash
# One eager bench run under rocprofv3 --kernel-trace; leaves the trace, the step sequence and the phase summary in gpurun_out/<tag>_*.
# usage: bash tools/trace_step.sh <tag> [bench args...]
TAG=${1:-trace}; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_$TAG
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$TAG -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-graph --no-fresh-input --no-cpu-baseline --no-kernel-table "$@" > /tmp/tr_$TAG.log 2>&1
t=$(find /tmp/tr_$TAG -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] || { echo "no trace"; tail -5 /tmp/tr_$TAG.log; exit 1; }
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python $GRAFT_REPO_ROOT/tools/step_sequence.py $t > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_sequence.txt
python $GRAFT_REPO_ROOT/tools/phase_trace.py $t > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_phases.txt
python - "$t" "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_laststep.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', ''), r['Grid_Size_X'], r['Workgroup_Size_X']) for r in rows), key=lambda t: t[0])
adam = [i for i, k in enumerate(ks) if 'adam_kernel' in k[2]]
step = ks[adam[-2] + 1:adam[-1] + 1]
t0 = step[0][0]
with open(sys.argv[2], 'w') as f:
    f.write('start_us,dur_us,queue,grid,wg,name\n')
    for s, e, n, q, g, w in step:
        f.write('%.1f,%.1f,%s,%s,%s,%s\n' % ((s - t0) / 1e3, (e - s) / 1e3, q, g, w, n.split('(')[0].replace('void ', '').replace('vq::', '')[:90].replace(',', ';')))
PY
head -12 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_phases.txt
