cd $GRAFT_REPO_ROOT
for v in VQVAE_X3_LEAN=0 VQVAE_X3_TAP2=0 VQVAE_X3_NB=3 X=1; do
  echo "== $v: $(env $v timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py -m gpu -x -q -k 'config4 or bf16' 2>&1 | tail -1)"
done
