cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "bf16 or config4" 2>&1 | tail -8
for g in 1 0; do echo "== VQVAE_R16=$g"; VQVAE_R16=$g bash tools/kstats.sh --workload c5 --bf16 --no-graph --no-fresh-input 2>&1 | sed -n 1,9p | cut -c1-150; done
