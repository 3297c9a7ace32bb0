cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_bench_shapes.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -8
