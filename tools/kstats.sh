#!/bin/bash
# per-kernel rocprofv3 stats of a short bench run -> gpurun_out/kstats.csv and a top-10 print
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > /tmp/b.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { echo "no stats"; tail -5 /tmp/b.log; exit 1; }
mkdir -p $GRAFT_REPO_ROOT/gpurun_out; cp $f $GRAFT_REPO_ROOT/gpurun_out/kstats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('sum of kernel time per step: %.2f ms' % (tot / 1e6 / 13))
for r in rows[:10]:
    print('%-62s calls/step %5.1f avg %8.1f us  ms/step %6.2f' % (r['Name'][:62], int(r['Calls']) / 13, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6 / 13))
PY
grep "^{" /tmp/b.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'gate TF', d['roofline']['achieved'])"
t=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
h = collections.defaultdict(list)
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    h[(r['Kernel_Name'][:34], r['Grid_Size_X'], r.get('Grid_Size_Y'))].append(d)
for k, v in sorted(h.items(), key=lambda kv: -sum(kv[1]))[:32]:
    print(k, 'n/step %.1f' % (len(v) / 13), 'avg %.1f us' % (sum(v) / len(v)), 'ms/step %.2f' % (sum(v) / 13 / 1e3))
PY
python - "$t" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'conv_gemm_kernel<0, 4' in r['Kernel_Name']]
d = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows)
h = collections.Counter(int(x // 20) * 20 for x in d)
print('conv_gemm<0,4> duration histogram (us bucket: launches/step):', {k: round(v / 13, 1) for k, v in sorted(h.items())})
PY
