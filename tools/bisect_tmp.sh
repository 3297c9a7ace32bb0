cd $GRAFT_REPO_ROOT
T="tests/test_gpu_model.py::test_training_reduces_the_losses"
for env in "A=1" "VQVAE_PACK_ONCE=0" "VQVAE_UPS_SEG=0" "VQVAE_X3_TAP2=0" "VQVAE_MATMUL=float32"; do
  echo "== $env"; env $env timeout 300 python -m pytest $T -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|Aborted|Error" | head -3
done
