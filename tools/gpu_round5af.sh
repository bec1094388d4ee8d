set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | cut -c1-200
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5af_bench.json 2> gpurun_out/r5af_bench.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r5af_bench.json')); print(d['ms_per_step'], d['ms_per_step_stats']['median'], d['train_loop_ms_per_step']); print({k:(v.get('ms_per_step'), v.get('error')) for k,v in d['other_configs'].items() if isinstance(v,dict)})"
