#!/bin/bash
# bench.py with the recorded step (hipGraph replay) under rocprofv3 --kernel-trace: the last replayed step's launches with
# their real gaps -> gpurun_out/<tag>_graph_laststep.csv and a sequence summary.  usage: bash tools/trace_graph.sh <tag> [bench args]
TAG=${1:-g}; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tg_$TAG
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tg_$TAG -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-fresh-input --no-cpu-baseline --no-kernel-table "$@" > /tmp/tg_$TAG.log 2>&1
t=$(find /tmp/tg_$TAG -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] || { echo "no trace"; tail -5 /tmp/tg_$TAG.log; exit 1; }
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
grep "^{" /tmp/tg_$TAG.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step under the profiler', d['ms_per_step'], d['step_execution'][:30])"
python - "$t" "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_graph_laststep.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', ''), r['Grid_Size_X'], r['Workgroup_Size_X']) for r in rows), key=lambda t: t[0])
adam = [i for i, k in enumerate(ks) if 'adam_kernel' in k[2]]
# the last TIMED replay: bench.py runs eager steps after the timed region (roofline pass) -- take the replay with the most regular spacing: the 4th from the end of the first 9+... simply the step before the last gap > 2 ms
# bench.py (steps = 6): ... 6 timed replays, 1 eager step, 6 eager steps of the roofline pass -> the 4th timed replay ends at adam[-10]
step = ks[adam[-11] + 1:adam[-10] + 1]
t0 = step[0][0]
print('%d launches, %.3f ms wall (first start to adam end), %.3f ms kernel time' % (len(step), (step[-1][1] - t0) / 1e6, sum(e - s for s, e, *_ in step) / 1e6))
with open(sys.argv[2], 'w') as f:
    f.write('start_us,end_us,dur_us,queue,grid,wg,name\n')
    for s, e, n, q, g, w in step:
        f.write('%.1f,%.1f,%.1f,%s,%s,%s,%s\n' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, g, w, n.split('(')[0].replace('void ', '').replace('vq::', '')[:90].replace(',', ';')))
PY
