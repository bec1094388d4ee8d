"""Latency of the inference forward (model.py:531-538; what Finetune/embed.py calls per request batch) at the cfg-3 model size:
wall time per `model(d, "retrieval" | "ranking")` call = batch upload + forward + output copy, by rows per request batch."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recommendersystem_amd as ra  # noqa: E402
from recommendersystem_amd import workload  # noqa: E402

cfg = workload.make_config(sys.argv[1] if len(sys.argv) > 1 else "cfg3")
S = cfg["max_sequence_length"]
model = ra.RecommenderModel(cfg, device=0, dtype="bf16", max_rows=16)
model.init_weights(0x1217); model.random_pretrained_embeddings(0x3E7A)
for rows in (1, 4, 16):
    d = workload.make_batch(cfg, rows, 5 + rows, mu=4.6, sigma=1.0)
    d["rope_input_pos"] = np.tile(np.arange(S, dtype=np.int32), rows)
    for task in ("retrieval", "ranking"):
        for _ in range(3):
            out = model(d, task)
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            out = model(d, task)
        dt = (time.perf_counter() - t0) / n
        print(f"rows {rows:2d} {task:9s}: {dt * 1e3:8.3f} ms per call   ({rows * S / dt / 1e3:8.1f} K interactions/s)  out {out.shape} finite {bool(np.isfinite(out).all())}", flush=True)
        # what a server reads (embed.py:147-161): one token per user for retrieval, 256 candidate tokens per user for ranking
        idx = np.concatenate([r * 2 * S + (np.array([2 * (S - 1)]) if task == "retrieval" else 2 * (S // 2 + np.arange(S // 2)) + 1) for r in range(rows)])
        for _ in range(3):
            out = model.inference_select(d, task, idx)
        t0 = time.perf_counter()
        for _ in range(n):
            out = model.inference_select(d, task, idx)
        dt = (time.perf_counter() - t0) / n
        print(f"rows {rows:2d} {task:9s}: {dt * 1e3:8.3f} ms per call   selected tokens only, out {out.shape}", flush=True)
