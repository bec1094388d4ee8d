# Where a dK/dV workgroup's time goes: builds attention.hip with -DATTN_KV_TRACE into tools/micro/bin/librsys_hip_trace.so (the product
# library is untouched) and runs tools/dbg/attn_kv_trace.py on the GPU box:  gpurun -- bash tools/trace_attn_kv.sh
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/recommendersystem_amd/csrc
mkdir -p $R/tools/micro/bin
[ -f $R/tools/micro/bin/librsys_hip_trace.so ] || {
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wno-unused-result -Wno-unused-value -I../../include -DATTN_KV_TRACE -c attention.hip -o /tmp/attention_trace.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/*.o | grep -v "attention.o\|h5_") /tmp/attention_trace.o -o $R/tools/micro/bin/librsys_hip_trace.so -ldl
}
cd $R && RSYS_LIB_PATH=$R/tools/micro/bin/librsys_hip_trace.so python tools/dbg/attn_kv_trace.py
