#!/bin/bash
# A/B of alternative builds of the library on one box: bench.py with each tools/experiments/abl/lib_<name>.so in turn
# usage: bash tools/ab_libs.sh "<bench args>" name1 name2 ...   ("product" = the committed build)
cd $GRAFT_REPO_ROOT
ARGS=$1; shift
cp chainer-vq-vae_amd/libvqvae_hip.so /tmp/lib_keep.so
for rep in 1 2; do
for n in "$@"; do
  if [ "$n" = product ]; then cp /tmp/lib_keep.so chainer-vq-vae_amd/libvqvae_hip.so; else cp tools/experiments/abl/lib_$n.so chainer-vq-vae_amd/libvqvae_hip.so; fi
  python bench.py --no-cpu-baseline $ARGS | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$n', round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms']*1e3,1), round(d['roofline']['frac'],4))"
done; done
cp /tmp/lib_keep.so chainer-vq-vae_amd/libvqvae_hip.so
