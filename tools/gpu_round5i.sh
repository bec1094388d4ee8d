set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 $R/tools/ab_loader_interval.py > $R/gpurun_out/r5i_loader_interval.log 2>&1; echo "rc=$?"; cat $R/gpurun_out/r5i_loader_interval.log | grep "rep"
timeout -k 10 200 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/r5i_bench.json 2> $R/gpurun_out/r5i_bench.err; python3 -c "
import json; d=json.load(open('$R/gpurun_out/r5i_bench.json')); print(d['ms_per_step'], d['ms_per_step_stats']['median'], d['train_loop_ms_per_step'])"
