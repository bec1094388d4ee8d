set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attention" > gpurun_out/r5b_attn_tests.log 2>&1; echo "attn tests rc=$?"; tail -4 gpurun_out/r5b_attn_tests.log
timeout -k 10 300 python -m pytest tests/test_gpu_prefetch.py tests/test_gpu_split_table_reduce.py -m gpu -q > gpurun_out/r5b_new_tests.log 2>&1; echo "new tests rc=$?"; tail -4 gpurun_out/r5b_new_tests.log
timeout -k 10 400 python -m pytest tests/test_gpu_bench_shape.py -m gpu -q -k "cfg5_bf16" > gpurun_out/r5b_cfg5.log 2>&1; echo "cfg5 rc=$?"; grep -h "cfg-5 LoRA step, \|passed\|failed" gpurun_out/r5b_cfg5.log | cut -c1-600
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in 0 1; do
export RSYS_ATTN_KV32=$v
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r5b_attn_${v}_$rep --output-format csv -- python3 $R/tools/bench_attn.py 8 > /dev/null 2>&1
echo "RSYS_ATTN_KV32=$v rep $rep"; grep -h "attn_" $R/gpurun_out/r5b_attn_${v}_$rep/*/*kernel_stats.csv | awk -F, '{print $1, $2, $4}' | cut -c1-140
done; done
