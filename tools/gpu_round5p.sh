set -o pipefail
cd $GRAFT_REPO_ROOT
for m in 0 1; do
echo "== RSYS_GEMM_PATCH=$m"
RSYS_GEMM_PATCH=$m timeout -k 10 60 ./tools/micro/bin/gemm4a 5 8192 8192 8192 2>&1 | grep "^M=" | cut -c1-200
for sh in "65536 2816 512 0 0" "4096 120000 512 0 0" "8192 8192 8192 0 0" "65536 5632 2048 0 0" "65536 2048 5632 0 0"; do
RSYS_GEMM_PATCH=$m timeout -k 10 120 python tools/bench_gemm.py $sh 2>&1 | grep "^M="
done
done
RSYS_GEMM_PATCH=2 timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
