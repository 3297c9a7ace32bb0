#!/bin/bash
# times the dilated-conv forward (256 x 256 tiles) with parts of the K loop compiled out (tools/experiments/abl/lib_abl<N>.so,
# built with `make EXTRA=-DX3_ABL=N`): 0 product, 1 no activation loads, 2 no weight loads, 3 no fragment reads, 4 no LDS writes, 5 = 1+2+4
cd $GRAFT_REPO_ROOT
cp chainer-vq-vae_amd/libvqvae_hip.so /tmp/lib_keep.so
for v in ${ABLS:-0 1 2 3 4 5}; do
  cp tools/experiments/abl/lib_abl$v.so chainer-vq-vae_amd/libvqvae_hip.so
  echo "== X3_ABL=$v"; VQVAE_X3_NB=3 python tools/occ_scaling.py 2>&1 | grep -E "^B  1 |^B 16"
done
cp /tmp/lib_keep.so chainer-vq-vae_amd/libvqvae_hip.so
