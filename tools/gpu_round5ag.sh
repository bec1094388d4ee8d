set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
t0=$(date +%s)
timeout -k 10 900 python bench.py > gpurun_out/r5ag_bench_default.json 2> gpurun_out/r5ag_bench_default.err; echo "default bench rc=$? in $(( $(date +%s) - t0 )) s"
python3 -c "
import json; d=json.load(open('gpurun_out/r5ag_bench_default.json')); print(d['steps'], d['warmup'], d['ms_per_step'], d['ms_per_step_stats']['median'], d['value'], d['roofline']['frac'], d['roofline']['instrumented_steps'])"
