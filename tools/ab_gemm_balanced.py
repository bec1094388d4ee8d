"""A/B of the balanced split-K mapping of the 128x128 weight-gradient kernel (RSYS_GEMM_BALANCED=0 / 1) on the step's dW shapes.
The env switch is read once per process: each arm runs in its own child process, arms alternate."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import bench_gemm as bg
NT = 65536
for (M, N, sk) in [(2816, 512, 8), (512, 1408, 16), (1024, 512, 16), (512, 512, 32)]:
    bg.run(M, N, NT, True, True, c_f32=True, splitk=sk, reps=20)
''' % HERE
for rep in range(2):
    for arm in ("0", "1"):
        env = dict(os.environ, RSYS_GEMM_BALANCED=arm, RSYS_GEMM_KERNEL_TN="1")
        print(f"--- RSYS_GEMM_BALANCED={arm} rep {rep}", flush=True)
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=True)
