"""The real input path: train_epoch fed from blosc HDF5 shard files (data.PretrainDataset: decode + block shuffle per file) with and
without the prefetch thread, against the in-memory loop of bench.py.  cfg-3, one GPU."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recommendersystem_amd as ra  # noqa: E402
from recommendersystem_amd import data, h5, workload  # noqa: E402
from recommendersystem_amd.train import ConstantScheduler, LambdaLR, train_epoch  # noqa: E402

cfg = workload.make_config("cfg3")
S, rows = cfg["max_sequence_length"], 64
tmp = tempfile.mkdtemp(dir="/tmp")
base = workload.make_stream(cfg, 262144, 1)
os.makedirs(f"{tmp}/training/1", exist_ok=True)
for p in range(5):
    d = {k: v.copy() for k, v in base.items()}
    d["userid"] = np.where(d["userid"] > 0, d["userid"] + p * 100000, 0).astype(np.int32)
    h5.write_h5(f"{tmp}/training/1/{p}.h5", d, blosc=3)
model = ra.RecommenderModel(cfg, device=0, dtype="bf16", max_rows=rows)
model.init_weights(0x1217); model.random_pretrained_embeddings(0x3E7A)
opt = ra.create_optimizer(model, cfg)
sched = LambdaLR(ConstantScheduler())
tw = ra.make_task_weights()
for name, wrap in (("synchronous loader", lambda ds: ds), ("prefetch thread", data.Prefetch), ("synchronous loader", lambda ds: ds), ("prefetch thread", data.Prefetch)):
    ds = wrap(data.PretrainDataset(f"{tmp}/training", 0, 1, rows * S, seed=3))
    ra.synchronize(); t0 = time.perf_counter()
    train_epoch(model, ds, opt, sched, tw, 1, None)
    ra.synchronize(); dt = time.perf_counter() - t0
    n = 5 * 262144 // (rows * S)
    print(f"{name:20s}: {n} steps, {dt / n * 1e3:7.2f} ms/step, {n * rows * S / dt / 1e6:6.3f} M interactions/s", flush=True)
