#!/bin/bash
# per-launch time of the ResidualBlock kernels with the gate epilogue's stores (X3_ABL=10) or the whole gate
# epilogue (11) compiled out (tools/experiments/abl/lib_abl<N>.so; wrong results, timing only)
cd $GRAFT_REPO_ROOT
cp chainer-vq-vae_amd/libvqvae_hip.so /tmp/lib_keep.so
echo "== product"; python tools/gbwd_scaling.py 64 2>&1 | grep -E "^B  1 |^B 16"
for v in ${ABLS:-10 11}; do
  cp tools/experiments/abl/lib_abl$v.so chainer-vq-vae_amd/libvqvae_hip.so
  echo "== X3_ABL=$v"; python tools/gbwd_scaling.py 64 2>&1 | grep -E "^B  1 |^B 16"
done
cp /tmp/lib_keep.so chainer-vq-vae_amd/libvqvae_hip.so
