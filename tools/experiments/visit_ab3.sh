# same-box A/B of library builds with the per-kernel table: bash tools/experiments/visit_ab3.sh name1 name2 ... (product = the in-tree build)
cd $GRAFT_REPO_ROOT; cp chainer-vq-vae_amd/libvqvae_hip.so /tmp/lib_keep.so
for rep in 1 2; do for n in "$@"; do
  if [ "$n" = product ]; then cp /tmp/lib_keep.so chainer-vq-vae_amd/libvqvae_hip.so; else cp tools/experiments/abl/lib_$n.so chainer-vq-vae_amd/libvqvae_hip.so; fi
  echo "== $n"; bash tools/quick_bench.sh 2>&1 | grep -E "ms/step|gz =|dilated conv \+|backward-data"
done; done
cp /tmp/lib_keep.so chainer-vq-vae_amd/libvqvae_hip.so
