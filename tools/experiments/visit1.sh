cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
./tools/ubench/gate_dma_probe > gpurun_out/gate_dma_probe.txt 2>&1; tail -40 gpurun_out/gate_dma_probe.txt | head -5
grep "^\[3\]" gpurun_out/gate_dma_probe.txt
cp chainer-vq-vae_amd/libvqvae_hip.so /tmp/lib_keep.so
for n in product perm; do
  if [ "$n" = product ]; then cp /tmp/lib_keep.so chainer-vq-vae_amd/libvqvae_hip.so; else cp tools/experiments/abl/lib_$n.so chainer-vq-vae_amd/libvqvae_hip.so; fi
  echo "=== $n"; bash tools/kstats.sh --no-graph 2>&1 | head -14; cp gpurun_out/kstats.csv gpurun_out/kstats_$n.csv
done
cp /tmp/lib_keep.so chainer-vq-vae_amd/libvqvae_hip.so
