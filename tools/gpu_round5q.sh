set -o pipefail
cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "== uniform"; timeout -k 10 60 ./tools/micro/bin/gemm4a 5 8192 8192 8192 2>&1 | grep "^M=" | cut -c1-200
echo "== normal"; G4_NORMAL=1 timeout -k 10 60 ./tools/micro/bin/gemm4a 5 8192 8192 8192 2>&1 | grep "^M=" | cut -c1-200
done
timeout -k 10 120 python tools/bench_gemm.py 8192 8192 8192 0 0 2>&1 | grep "^M="
python - <<'PY'
import torch
A=torch.randn(8192,8192,device="cuda",dtype=torch.bfloat16); B=torch.randn(8192,8192,device="cuda",dtype=torch.bfloat16)
U=(torch.rand(8192,8192,device="cuda")*2-1).to(torch.bfloat16); V=(torch.rand(8192,8192,device="cuda")*2-1).to(torch.bfloat16)
def t(a,b):
    for _ in range(3): torch.mm(a,b.t())
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): torch.mm(a,b.t())
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/10*1e3
for i in range(2):
    print("vendor 8192^3 normal %.1f us   uniform(-1,1) %.1f us" % (t(A,B), t(U,V)))
PY
