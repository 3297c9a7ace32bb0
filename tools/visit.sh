cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --no-header -p no:cacheprovider -k "float32x3_is or scale_invariant or conv1d_fwd_bwd or resblock or random_shapes" 2>&1 | tail -15
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --matmul float32x2 2>&1 | tail -3 | cut -c1-600
