set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
for r in 2 4 8 11; do
echo "== rows $r"
for sh in "65536 2816 512 0 0" "65536 5632 2048 0 0" "8192 8192 8192 0 0"; do
RSYS_GEMM_PATCH_ROWS=$r timeout -k 10 120 python tools/bench_gemm.py $sh 2>&1 | grep "^M="
done
done
for r in 2 8; do
bash tools/ab_env.sh RSYS_GEMM_PATCH_ROWS $r 1 2>&1 | cut -c1-120
done
