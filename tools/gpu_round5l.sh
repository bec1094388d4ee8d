set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
for v in "" _nodma _nobarrier _nodma_nobarrier "" _nodma_nobarrier; do
echo "== variant '$v'"; timeout -k 10 60 ./tools/micro/bin/gemm4a$v 5 4096 4096 8192 2>&1 | grep "^M=" | cut -c1-150
done
