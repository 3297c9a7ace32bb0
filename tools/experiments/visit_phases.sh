cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp chainer-vq-vae_amd/libvqvae_hip.so /tmp/lib_keep.so
cp tools/experiments/abl/lib_phase.so chainer-vq-vae_amd/libvqvae_hip.so
python tools/experiments/phases.py 2>/dev/null | grep workgroups
cp /tmp/lib_keep.so chainer-vq-vae_amd/libvqvae_hip.so
