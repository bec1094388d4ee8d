set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/r5w_bench_$i.json 2> gpurun_out/r5w_bench_$i.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r5w_bench_$i.json')); print(d['ms_per_step'], d['ms_per_step_stats']['median'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['launches'], d['roofline']['share_of_step'], d['per_kernel_fields_measured_on'][:40]); print(d['ms_per_step_by_phase']); print(d['gemm_variants']['8c'])"
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --detail --no-cpu-baseline --no-train-loop 2>&1 >/dev/null | head -8
