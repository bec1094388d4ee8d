set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out tools/micro/bin
(cd recommendersystem_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wno-unused-result -Wno-unused-value -I../../include -DATTN_KV32_WPS=3 -c attention.hip -o /tmp/attention_w3.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/*.o | grep -v "attention.o\|h5_") /tmp/attention_w3.o -o $R/tools/micro/bin/librsys_hip_kv32w3.so -ldl) || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attention" > gpurun_out/r5e_attn_tests.log 2>&1; echo "attn tests rc=$?"; tail -3 gpurun_out/r5e_attn_tests.log
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in "w2 1" "w3 1" "w2 0"; do
set -- $v
export RSYS_ATTN_KV32=$2
if [ $1 = w3 ]; then export RSYS_LIB_PATH=$R/tools/micro/bin/librsys_hip_kv32w3.so; else unset RSYS_LIB_PATH; fi
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r5e_attn_${1}_${2}_$rep --output-format csv -- python3 $R/tools/bench_attn.py 8 > /dev/null 2>&1
echo "build=$1 KV32=$2 rep $rep"; grep -h "attn_bwd_kv" $R/gpurun_out/r5e_attn_${1}_${2}_$rep/*/*kernel_stats.csv | awk -F, '{print $1, $2, $4}' | cut -c1-140
done; done
