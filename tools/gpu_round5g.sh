set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "gemm" > gpurun_out/r5g_gemm_tests.log 2>&1; echo "gemm tests rc=$?"; tail -4 gpurun_out/r5g_gemm_tests.log
timeout -k 10 900 python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_model.py tests/test_gpu_deterministic.py -m gpu -q -x > gpurun_out/r5g_model_tests.log 2>&1; echo "model tests rc=$?"; tail -4 gpurun_out/r5g_model_tests.log
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
if [ $v = 0 ]; then export RSYS_GEMM_KERNEL_TN=1; else unset RSYS_GEMM_KERNEL_TN; fi
timeout -k 10 300 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra-legs --no-train-loop > $R/gpurun_out/r5g_bench_$v.json 2> $R/gpurun_out/r5g_bench_$v.err; echo "bench $v rc=$?"; python3 -c "
import json; d=json.load(open('$R/gpurun_out/r5g_bench_$v.json')); print(d['ms_per_step'], d['ms_per_step_stats']['median'], d['ms_per_step_by_phase']['phase_heads'], d['gemm_variants'])"
done
unset RSYS_GEMM_KERNEL_TN
timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r5g_bench_full.json 2> $R/gpurun_out/r5g_bench_full.err; python3 -c "
import json; d=json.load(open('$R/gpurun_out/r5g_bench_full.json')); print(d['ms_per_step'], d['ms_per_step_stats']['median'], d['train_loop_ms_per_step'], d['other_configs']['hdf5_loop_cfg3'])"
