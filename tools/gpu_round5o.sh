set -o pipefail
cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout -k 10 60 ./tools/micro/bin/gemm4a 5 8192 8192 8192 2>&1 | grep -A1 "^M=" | cut -c1-200
timeout -k 10 60 ./tools/micro/bin/gemm4a 5 65536 512 2816 2>&1 | grep -A1 "^M=" | cut -c1-200
timeout -k 10 60 ./tools/micro/bin/gemm4a 5 65536 512 1408 2>&1 | grep -A1 "^M=" | cut -c1-200
done
