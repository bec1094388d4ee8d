import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import recommendersystem_amd as ra
from recommendersystem_amd import workload
for dtype in ("fp8", "bf16", "fp8", "bf16"):
    cfg = workload.make_config("cfg3", learning_rate=3e-4)
    rows = 64
    model = ra.RecommenderModel(cfg, device=0, dtype=dtype, max_rows=rows)
    model.init_weights(0x1217); model.random_pretrained_embeddings(0x3E7A)
    opt = ra.create_optimizer(model, cfg)
    model.set_loss_weights(ra.make_task_weights(), 1)
    batches = [workload.make_batch(cfg, rows, 100 + i, mu=4.6, sigma=1.0) for i in range(8)]
    for mode in ("cycle8", "same"):
        for s in range(10):
            model(batches[s % 8 if mode == "cycle8" else 0], False); opt.step(lr_factor=1.0, clip_max_norm=1.0)
        t0 = time.time()
        for s in range(200):
            losses = model(batches[s % 8 if mode == "cycle8" else 0], False); opt.step(lr_factor=1.0, clip_max_norm=1.0)
        print(dtype, mode, round((time.time() - t0) / 200 * 1e3, 3), "ms/step", flush=True)
    model.close()
