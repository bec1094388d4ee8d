"""Worker of tests/test_gpu_bench_shape.py: the benchmarked step (cfg-3, 64 rows, bf16, initialisation-scale weights) under whatever
RSYS_* switches the parent set (they are read once per process); losses and the last layer's gradients to an .npz.
RSYS_TEST_CFG names another configuration, RSYS_TEST_DETERMINISTIC=1 runs the bitwise reproducible mode, RSYS_TEST_ALL_GRADS=1 keeps
every named gradient and the dense trunk output."""
import os
import sys

import numpy as np

sys.path.insert(0, sys.argv[2])
import recommendersystem_amd as ra  # noqa: E402
from oracle import synth, train_np  # noqa: E402

cfg = synth.make_config(os.environ.get("RSYS_TEST_CFG", "cfg3"), deterministic=os.environ.get("RSYS_TEST_DETERMINISTIC") == "1")
rows = 64
d = synth.make_batch(cfg, rows, 0xD47A, mu=4.6, sigma=1.0)
wm, rm = synth.make_masks(cfg, rows, 0x3A5C)
model = ra.RecommenderModel(cfg, dtype="bf16", max_rows=rows)
model.init_weights(0x1217)
model.random_pretrained_embeddings(0x3E7A)
rng = np.random.default_rng(3)          # (= _perturb_scales(model, 3) of the test module)
for n, shape, tr in model.named_parameters():
    if n.endswith(".scale") or "periodic_time" in n:
        model.set_parameter(n, (1.0 if n.endswith(".scale") else 0.0) + 0.1 * rng.standard_normal(shape).astype(np.float32))
model.set_loss_weights(train_np.make_task_weights(), 1)
losses = model(d, False, masks=(wm, rm))
L = cfg["num_layers"] - 1
res = {"losses": np.array(losses, np.float64)}
every = os.environ.get("RSYS_TEST_ALL_GRADS") == "1"
for n in synth.trainable_names(cfg):
    if every or n.startswith(f"transformers.layers.{L}.") or n.startswith(f"transformers.layers.{L - 1}.attn."):
        res["g/" + n] = model.grad(n)
if every:
    res["trunk"] = model.trunk_output(rows)
np.savez(sys.argv[1], **res)
model.close()
