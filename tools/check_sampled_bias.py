"""Bias check of the sampled soft-max on a PEAKED model: train the full-soft-max model for a while, then compare the full gradient of
one batch with the mean of many sampled-soft-max gradients (fresh samples) of the same batch and weights."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recommendersystem_amd as ra  # noqa: E402
from recommendersystem_amd import train as rtrain, workload  # noqa: E402
from tools.converge_sampled import masks_for  # noqa: E402

E = "item_embedding.matchedid_embedding.embedding.weight"
cfg = workload.make_config("cfg2", num_layers=2, learning_rate=1e-3)
rows, S = 32, cfg["max_sequence_length"]
stream = workload.make_stream(cfg, 21 * rows * S, 0xD47A)
batches = [{k: v[i * rows * S:(i + 1) * rows * S] for k, v in stream.items()} for i in range(21)]
tw = rtrain.make_task_weights()
full = ra.RecommenderModel(dict(cfg, table_shard=(0, 1)), device=0, dtype="bf16", max_rows=rows)
full.random_pretrained_embeddings(0x3E7A); full.init_weights(0x1217)
opt = ra.create_optimizer(full, cfg)
full.set_loss_weights(tw, 1)
for step in range(200):
    l = full(batches[step % 20], False, masks=masks_for(cfg, rows, 100 + step))
    opt.step(lr_factor=min(1.0, (step + 1) / 50.0), clip_max_norm=1.0)
print("trained, losses", l)
b = batches[20]; mk = masks_for(cfg, rows, 7)
full.zero_grad()
l_full = full(b, False, masks=mk)
g_full = {n: full.grad(n).astype(np.float64) for n in (E, "transformers.layers.0.mlp.w1.weight", "item_embedding.projection_layer.bias")}
P = {k: v for k, v in full.state_dict().items() if not k.startswith("watch_head.")}
for neg in (4096, 16384):
    samp = ra.RecommenderModel(dict(cfg, table_shard=(0, 1), sampled_softmax=neg), device=0, dtype="bf16", max_rows=rows)
    samp.random_pretrained_embeddings(0x3E7A)
    samp.load_state_dict(P)
    samp.set_loss_weights(tw, 1)
    samp.upload(b, mk)
    n_rep = 48
    acc = {n: 0 for n in g_full}; ls = []
    for r in range(n_rep):
        samp.zero_grad()
        samp.forward_resident(False, step=1000 + r)
        ls.append(samp.losses(False))
        for n in g_full:
            acc[n] = acc[n] + samp.grad(n).astype(np.float64)
    ls = np.array(ls)
    print(f"neg={neg}: loss full {l_full[0]:.4f} {l_full[2]:.4f}  sampled mean {ls[:, 0].mean():.4f} {ls[:, 2].mean():.4f} std {ls[:, 0].std():.4f} {ls[:, 2].std():.4f}")
    for n in g_full:
        m = acc[n] / n_rep
        num = np.abs(m - g_full[n]).sum(); den = np.abs(g_full[n]).sum()
        cos = (m * g_full[n]).sum() / (np.linalg.norm(m) * np.linalg.norm(g_full[n]) + 1e-30)
        print(f"   {n[:50]:50s} |mean - full|_1 / |full|_1 = {num / den:.3f}  cos {cos:.4f}  norm ratio {np.linalg.norm(m) / np.linalg.norm(g_full[n]):.3f}")
    gE, mE = g_full[E], acc[E] / n_rep
    V0 = cfg["vocab_sizes"]["0_matchedid"]
    for name, lo in (("manga", 0), ("anime", V0)):
        top = lo + np.arange(1, 9)
        print(f"   {name} top rows |g| full  ", np.round(np.linalg.norm(gE[top], axis=1), 5))
        print(f"   {name} top rows |g| sampled", np.round(np.linalg.norm(mE[top], axis=1), 5))
        print(f"   {name} top rows cos        ", np.round([(gE[i] * mE[i]).sum() / (np.linalg.norm(gE[i]) * np.linalg.norm(mE[i]) + 1e-30) for i in top], 3))
    samp.close()
