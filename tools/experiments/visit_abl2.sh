# kernel-stats A/B of the working tree's library against tools/experiments/abl/$1 (same visit, interleaved)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
if [ -n "$2" ]; then timeout 2400 python -m pytest tests -m gpu -x -q --no-header -p no:cacheprovider -k "$2" 2>&1 | tail -4; fi
cp chainer-vq-vae_amd/libvqvae_hip.so /tmp/lib_keep.so
for i in 1 2; do
for lib in tools/experiments/abl/$1 /tmp/lib_keep.so; do
cp $lib chainer-vq-vae_amd/libvqvae_hip.so
rm -rf /tmp/st
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o st -- python bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline > /tmp/bench.log 2>&1
f=$(find /tmp/st -name "*kernel_stats.csv" | head -1)
echo "== $lib  $(tail -1 /tmp/bench.log | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.3f ms/step (under rocprof)" % j["ms_per_step"])')"; python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:7]:
    print('  %4d %8.1f us  %s'%(int(r['Calls']), float(r['AverageNs'])/1e3, r['Name'][:80]))
P
done; done
cp /tmp/lib_keep.so chainer-vq-vae_amd/libvqvae_hip.so
for i in 1 2; do for lib in tools/experiments/abl/$1 /tmp/lib_keep.so; do
cp $lib chainer-vq-vae_amd/libvqvae_hip.so
echo "$lib: $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.3f ms  gate launch %.2f us" % (j["ms_per_step"], 1e3*j["roofline"]["avg_launch_ms"]))')"
done; done
cp /tmp/lib_keep.so chainer-vq-vae_amd/libvqvae_hip.so
