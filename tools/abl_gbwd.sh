#!/bin/bash
# per-launch time of the ResidualBlock kernels (gate-derivative GEMM = the non-lean conv_gemm_x3_kernel loop) with parts
# of that loop compiled out: X3_ABL 1 no activation loads, 2 no weight loads, 3 no fragment reads, 5 = 1 + 2 + no LDS writes
cd $GRAFT_REPO_ROOT
cp chainer-vq-vae_amd/libvqvae_hip.so /tmp/lib_keep.so
echo "== product"; python tools/gbwd_scaling.py 64 2>&1 | grep -E "^B  1 |^B 16"
for v in ${ABLS:-1 2 3 5}; do
  cp tools/experiments/abl/lib_abl$v.so chainer-vq-vae_amd/libvqvae_hip.so
  echo "== X3_ABL=$v"; python tools/gbwd_scaling.py 64 2>&1 | grep -E "^B  1 |^B 16"
done
cp /tmp/lib_keep.so chainer-vq-vae_amd/libvqvae_hip.so
