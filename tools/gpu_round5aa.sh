set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
t0=$(date +%s)
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r5aa_bench.json 2> gpurun_out/r5aa_bench.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"
python3 -c "
import json; d=json.load(open('gpurun_out/r5aa_bench.json')); print(d['ms_per_step'], d['ms_per_step_stats']['median']); print(json.dumps(d['roofline'])[:900]); print(d['hbm_kernels']['rmsnorm_bwd'])"
tail -3 gpurun_out/r5aa_bench.err
