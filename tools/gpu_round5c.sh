set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_bench_shape.py -m gpu -q -k "cfg5_bf16" > gpurun_out/r5c_cfg5.log 2>&1; echo "cfg5 rc=$?"; grep -h "cfg-5 LoRA step, \|passed\|failed" gpurun_out/r5c_cfg5.log | cut -c1-600
rm -f tools/micro/bin/librsys_hip_trace.so
RSYS_ATTN_KV32=1 bash tools/trace_attn_kv.sh > gpurun_out/r5c_trace_kv32.log 2>&1; echo "trace kv32 rc=$?"; cat gpurun_out/r5c_trace_kv32.log
RSYS_ATTN_KV32=0 bash tools/trace_attn_kv.sh > gpurun_out/r5c_trace_kv16.log 2>&1; echo "trace kv16 rc=$?"; cat gpurun_out/r5c_trace_kv16.log
