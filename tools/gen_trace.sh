#!/bin/bash
# kernel durations / gaps of the eager generation loop (rocprofv3 cannot trace the hipGraph replay)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o gen -- python $GRAFT_REPO_ROOT/tools/gen_bench.py --steps 1500 --graph-steps 0 --per-step-kernels "$@" > /tmp/log.txt 2>&1
grep workload /tmp/log.txt
t=$(find /tmp/prof -name "*kernel_trace.csv" 2>/dev/null | head -1)
[ -n "$t" ] || { echo "no trace"; exit 1; }
python - "$t" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("vq::gen")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = 0
for i, r in enumerate(rows):
    if 'finish' in r["Kernel_Name"]:
        per = i + 1
        break
seg = rows[per * 1000: per * 1000 + per]
prev = None
tot_d = tot_g = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = (s - prev) / 1e3 if prev else 0
    tot_d += (e - s) / 1e3; tot_g += g
    print(r["Kernel_Name"][:24], "dur %5.2f us" % ((e - s) / 1e3), "gap %5.2f us" % g, 'wg', r.get("Workgroup_Size_X"), 'grid', r.get("Grid_Size_X"))
    prev = e
print('kernels/step', per, 'sum dur %.1f us, sum gaps %.1f us' % (tot_d, tot_g))
PY
