set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
if [ $v = 0 ]; then export RSYS_GEMM_KERNEL_TN=1; else unset RSYS_GEMM_KERNEL_TN; fi
timeout -k 10 400 python3 $R/bench.py --config prod --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs --no-train-loop > $R/gpurun_out/r5h_prod_$v.json 2> $R/gpurun_out/r5h_prod_$v.err; echo "prod $v rc=$?"; python3 -c "
import json; d=json.load(open('$R/gpurun_out/r5h_prod_$v.json')); print(d['ms_per_step'], d['ms_per_step_by_phase']['phase_heads'], {k:v for k,v in d['gemm_variants'].items() if k in ('tn','8ts','nn')})"
done
unset RSYS_GEMM_KERNEL_TN
timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r5h_bench_full.json 2> $R/gpurun_out/r5h_bench_full.err; python3 -c "
import json; d=json.load(open('$R/gpurun_out/r5h_bench_full.json')); print(d['ms_per_step'], d['ms_per_step_stats']['median'], d['train_loop_ms_per_step'], d['other_configs']['hdf5_loop_cfg3'])"
