set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 60 ./tools/micro/bin/gemm4a_mfma32 2 2048 2048 1024 2>&1 | cut -c1-200 || exit 1
for v in _mfma32 "" _mfma32_nodma _mfma32 ""; do
echo "== variant '$v'"; timeout -k 10 60 ./tools/micro/bin/gemm4a$v 5 4096 4096 8192 2>&1 | grep -A1 "^M=" | cut -c1-200
done
echo "== 8192^3"; timeout -k 10 60 ./tools/micro/bin/gemm4a_mfma32 5 8192 8192 8192 2>&1 | grep -A1 "^M=" | cut -c1-200
