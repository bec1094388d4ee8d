set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 120 ./tools/micro/bin/gemm4a 5 > gpurun_out/r5j_gemm4a.log 2>&1; echo "rc=$?"; cat gpurun_out/r5j_gemm4a.log
