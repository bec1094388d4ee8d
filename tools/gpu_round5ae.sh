set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_bench_shape.py -x -q -m gpu -k "fused_item_table or cpp_oracle" 2>&1 | tail -8 | cut -c1-200
bash tools/ab_env.sh RSYS_TABLE_TAIL 0 2 2>&1 | cut -c1-100
for v in 1 0; do
RSYS_TABLE_TAIL=$v timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --detail --no-cpu-baseline --no-train-loop 2>&1 >/dev/null | grep -E "table|phase_embed " | cut -c1-100
done
