set -o pipefail
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for arm in r4 r5; do
if [ $arm = r4 ]; then D=$R/tools/dbg/r4tree; else D=$R; fi
cd $D && timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --detail --no-cpu-baseline --no-train-loop > /tmp/ab_$arm.json 2> /tmp/ab_$arm.txt; echo "$arm rc=$?"
python3 - <<PY
import json
d=json.loads(open('/tmp/ab_$arm.json').read().strip().splitlines()[-1])
print('$arm rep $rep', d['ms_per_step'], d.get('ms_per_step_stats',{}).get('median'))
PY
grep -E "gemm_dw_group|gemm_w13_fwd|gemm_w2_dx|gemm_w13_dx|attn_bwd|attn_fwd|hbm_rmsnorm_bwd|adamw|gemm_table|gemm_qkv_fwd@8c|gemm_o_fwd|gemm_w2_fwd|gemm_qkv_dx|gemm_head" /tmp/ab_$arm.txt | awk '{printf "%s %s | ", $1, $2}'; echo
done
done
