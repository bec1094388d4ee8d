set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
bash tools/ab_env.sh RSYS_GEMM_PATCH 0 2 2>&1 | cut -c1-260
for v in 1 0; do
if [ $v = 0 ]; then export RSYS_GEMM_PATCH=0; else unset RSYS_GEMM_PATCH; fi
timeout -k 10 400 python3 $R/bench.py --config prod --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs --no-train-loop > $R/gpurun_out/r5r_prod_$v.json 2> $R/gpurun_out/r5r_prod_$v.err; echo "prod patch=$v rc=$?"; python3 -c "
import json; d=json.load(open('$R/gpurun_out/r5r_prod_$v.json')); print(d['ms_per_step'], {k:v for k,v in d['gemm_variants'].items() if k in ('8c',)})"
done
