set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_prefetch.py tests/test_gpu_training_curve.py -x -q -m gpu 2>&1 | tail -5
