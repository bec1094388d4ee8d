set -o pipefail
cd $GRAFT_REPO_ROOT
bash tools/prof_round.sh r5zz
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python tools/soak.py 3000 > gpurun_out/r5zz_soak.log 2>&1; echo "soak rc=$?"; tail -4 gpurun_out/r5zz_soak.log
