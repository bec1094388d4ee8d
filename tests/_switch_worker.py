"""Worker of tests/test_gpu_switches.py: one optimizer step of a small model under whatever RSYS_* switches the parent set (they are
read once per process), results to an .npz."""
import sys

import numpy as np

sys.path.insert(0, sys.argv[3])
import recommendersystem_amd as ra  # noqa: E402
from oracle import synth  # noqa: E402

out, dtype = sys.argv[1], sys.argv[2]
import os  # noqa: E402
cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16, deterministic=os.environ.get("RSYS_TEST_DETERMINISTIC") == "1")
rows, seed = 4, 31
P = synth.make_params(cfg, seed, "test")
d = synth.make_batch(cfg, rows, seed + 1)
wm, rm = synth.make_masks(cfg, rows, seed + 2)
model = ra.RecommenderModel(cfg, dtype=dtype, max_rows=rows)
model.load_state_dict(P)
model.set_loss_weights([0.05, 0.2, 0.3, 0.25], 1)
losses = model(d, False, masks=(wm, rm))
names = synth.trainable_names(cfg)
res = {"losses": np.array(losses, np.float64), "trunk": model.trunk_output(rows)}
for n in names:
    res["g/" + n] = model.grad(n)
opt = ra.create_optimizer(model, dict(cfg, learning_rate=1e-4))
opt.step(clip_max_norm=1.0)
for n in names:
    res["p/" + n] = model.get_parameter(n)
np.savez(out, **res)
model.close()
