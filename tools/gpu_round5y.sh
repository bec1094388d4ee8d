set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_prefetch.py -x -q -m gpu 2>&1 | tail -5
for i in 1 2; do
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5y_bench_$i.json 2> gpurun_out/r5y_bench_$i.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r5y_bench_$i.json')); print(d['ms_per_step'], d['ms_per_step_stats']['median'], d['train_loop_ms_per_step'], d['other_configs']['hdf5_loop_cfg3'])"
done
