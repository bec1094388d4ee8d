"""Host CPU probe for the CPU-baseline leg: what the box lets this process use, and the SGEMM rate of oracle/cpu_step.cpp by thread count."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_step  # noqa: E402

for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(p, open(p).read().strip())
    except OSError as e:
        print(p, "-", e)
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count(), "host_cpus()", cpu_step.host_cpus())
rng = np.random.default_rng(0)
A = rng.standard_normal((4096, 4096), dtype=np.float32)
L = cpu_step.lib()
for n in (8, 16, 24, 32, 48, 64, 128):
    L.cpu_step_set_threads(n)
    best = 0
    for _ in range(3):
        t = time.time(); cpu_step.sgemm_nt(A, A); best = max(best, 2 * 4096 ** 3 / (time.time() - t))
    print(f"threads {n:4d}: {best / 1e9:8.0f} GFLOP/s", flush=True)
