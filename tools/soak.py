"""Soak: N optimizer steps at cfg-3 (bf16, fresh device-drawn masks each step, a few cycling batches, WSD warm-up then constant lr):
losses must stay finite and fall, device memory must not grow."""
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recommendersystem_amd as ra  # noqa: E402
from recommendersystem_amd import workload  # noqa: E402
from recommendersystem_amd.train import LambdaLR, WSDScheduler  # noqa: E402


def vram():
    out = subprocess.run(["rocm-smi", "--showmeminfo", "vram"], capture_output=True, text=True).stdout
    for line in out.splitlines():
        if "Used" in line:
            return int(line.split(":")[-1]) / 2 ** 30
    return float("nan")


steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"   # "fp8": the reference's float8 trunk
cfg = workload.make_config("cfg3", learning_rate=3e-4)
rows = 64
model = ra.RecommenderModel(cfg, device=0, dtype=dtype, max_rows=rows)
model.init_weights(0x1217); model.random_pretrained_embeddings(0x3E7A)
opt = ra.create_optimizer(model, cfg)
sched = LambdaLR(WSDScheduler(warmup_steps=200, total_steps=10 * steps, decay_ratio=0.1, final_ratio=0.1))
model.set_loss_weights(ra.make_task_weights(), 1)
batches = [workload.make_batch(cfg, rows, 100 + i, mu=4.6, sigma=1.0) for i in range(8)]
m0 = vram(); t0 = time.time()
hist = []
for s in range(steps):
    losses = model(batches[s % 8], False)
    opt.step(lr_factor=sched.factor(), clip_max_norm=1.0)
    sched.step()
    assert all(np.isfinite(losses)), (s, losses)
    if s % 200 == 0 or s == steps - 1:
        hist.append((s, [round(float(x), 4) for x in losses]))
        print(f"step {s:5d} losses {hist[-1][1]} lr factor {sched.factor():.3f} vram {vram():.2f} GiB", flush=True)
m1 = vram()
print(f"{dtype}: {steps} steps in {time.time() - t0:.1f}s; vram {m0:.2f} -> {m1:.2f} GiB")
assert hist[-1][1][0] < hist[0][1][0] - 1.0 and hist[-1][1][2] < hist[0][1][2] - 1.0, "watch losses did not fall"
assert abs(m1 - m0) < 0.25, "device memory grew"
print("ok")
