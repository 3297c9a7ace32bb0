cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sig; export TMPDIR=/tmp
for v in 3 7 3 7; do
rm -rf /tmp/st
VQVAE_PRESPLIT=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o st -- python bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline > /dev/null 2>&1
f=$(find /tmp/st -name "*kernel_stats.csv" | head -1)
echo "== PRESPLIT=$v"; python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time per step %.3f ms'%(tot/13e6))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:7]:
    print('  %4d %8.1f us  %s'%(int(r['Calls']), float(r['AverageNs'])/1e3, r['Name'][:80]))
P
done
