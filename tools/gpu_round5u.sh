set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5u_tests.log; rc=$?; tail -5 gpurun_out/r5u_tests.log; echo "pytest rc=$rc"
[ $rc = 0 ] && timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r5u_bench.json 2> gpurun_out/r5u_bench.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r5u_bench.json')); print(d['ms_per_step'], d['ms_per_step_stats']['median'], d['switches'], d['train_loop_ms_per_step'])"
