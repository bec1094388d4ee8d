set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_zero1.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -15
