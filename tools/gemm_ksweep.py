import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_gemm import run
for K in (64, 128, 256, 512, 1024, 2048):
    run(65536, 1024, K, False, False)
for K in (64, 512, 2048):
    run(65536, 512, K, False, False, c_f32=True)
