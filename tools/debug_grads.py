"""Debug helper (GPU): per-parameter gradient error of the HIP path against the numpy oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recommendersystem_amd as ra
from oracle import model_np, synth

name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp32"
over, rows, seed = {"tiny": (dict(mask_rate=0.25, mask_topk=6), 3, 11), "hd64": (dict(mask_rate=0.2, mask_topk=16), 2, 23)}[name]
cfg = synth.make_config(name, **over)
P = synth.make_params(cfg, seed, "test")
d = synth.make_batch(cfg, rows, seed + 1)
wm, rm = synth.make_masks(cfg, rows, seed + 2)
tw = [0.05, 0.2, 0.3, 0.25]
ref = model_np.OracleModel(cfg, P)
dm = model_np.mask_tokens(cfg, model_np.reshape_batch(cfg, d), wm, rm)
y_ref, _ = ref.embed(dm)
l_ref, G = ref.forward(dm, False, True, tw)
m = ra.RecommenderModel(cfg, dtype=dtype, max_rows=rows)
m.load_state_dict(P)
m.set_loss_weights(tw, 1)
l = m(d, False, masks=(wm, rm))
print("losses", l, l_ref)
y = m.trunk_output(rows)
print("trunk relerr", np.abs(y - y_ref).max() / np.abs(y_ref).max())
for n in synth.trainable_names(cfg):
    g = m.grad(n)
    e = np.abs(g - G[n]).max() / max(np.abs(G[n]).max(), 1e-12)
    print(f"{e:10.3e} {np.abs(G[n]).max():10.3e} {n}")
