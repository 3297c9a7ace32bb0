cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_kernels.py -m gpu -q -x --no-header -p no:cacheprovider -k "resstack or resblock or presplit or two_tap or conv1d" 2>&1 | tail -4
bash tools/ab_libs.sh "--steps 20 --warmup 5" product noadma
