cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp chainer-vq-vae_amd/libvqvae_hip.so /tmp/lib_keep.so
for i in 1 2; do
for lib in /tmp/lib_keep.so tools/experiments/abl/$1; do
  cp $lib chainer-vq-vae_amd/libvqvae_hip.so
  echo "$lib: $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.3f ms  gate launch %.2f us" % (j["ms_per_step"], 1e3*j["roofline"]["avg_launch_ms"]))')"
done; done
cp /tmp/lib_keep.so chainer-vq-vae_amd/libvqvae_hip.so
