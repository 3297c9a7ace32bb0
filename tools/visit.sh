cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
timeout 300 rocprofv3 --memory-copy-trace --hip-runtime-trace --stats --output-format csv -d /tmp/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-graph --no-fresh-input --workload c5 --bf16 > /tmp/b.log 2>&1
ls /tmp/prof/*/ 2>/dev/null | head; f=$(find /tmp/prof -name "*memory_copy_stats.csv" | head -1); cat $f | head
g=$(find /tmp/prof -name "*hip_api_stats.csv" | head -1); head -12 $g | cut -c1-120
t=$(find /tmp/prof -name "*memory_copy_trace.csv" | head -1); python - "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), rows[0].keys() if rows else None)
c = collections.Counter((r.get('Direction'), r.get('Bytes') or r.get('Size')) for r in rows)
for k, v in c.most_common(15): print(v, k)
PY
