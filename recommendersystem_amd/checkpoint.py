"""Checkpoint interchange with the reference's `transformer.masked.pt`.

The reference writes a torch pickle (transformer.py:456-466):
    {"model": state_dict, "optimizer": AdamW.state_dict(), "scheduler": LambdaLR.state_dict(),
     "config", "epoch", "training_loss", "test_loss"}
(finetune checkpoints carry only model / config / epoch / losses, transformer.py:468-477) and resumes from it
(transformer.py:657-665, 690-700).  This package checkpoints to a flat `.npz` (train.checkpoint_model): the
reference's state-dict names under "model/", AdamW moments by parameter NAME, the scheduler's step counter.

    python -m recommendersystem_amd.checkpoint pt2npz transformer.masked.pt transformer.masked.npz
    python -m recommendersystem_amd.checkpoint npz2pt transformer.masked.npz transformer.masked.pt [DATADIR | media_embeddings.h5]
(the optional last argument supplies the frozen metadata table the reference's strict load expects)

The conversion needs torch only to read / write the pickle (host side; nothing here touches the GPU path).
Optimizer state is keyed by parameter INDEX in the reference: `create_optimizer` (transformer.py:285-298) puts the
trainable parameters with dim >= 2 first (weight decay 0.1), then the others, each in `named_parameters()` order,
which is the state-dict order without the `watch_head.` aliases (shared with `item_embedding.`, model.py:354) and
without the frozen metadata table (model.py:113-114).
"""
import json
import sys

import numpy as np

FROZEN = "item_embedding.metadata_embedding.embedding.weight"
ALIAS = "watch_head."


def trainable_order(model_keys, shapes):
    """[(name, decay?)] in the reference optimizer's parameter-index order."""
    names = [k for k in model_keys if not k.startswith(ALIAS) and k != FROZEN]
    decay = [n for n in names if len(shapes[n]) >= 2]
    nodecay = [n for n in names if len(shapes[n]) < 2]
    return decay, nodecay


def from_reference(ckpt):
    """reference checkpoint dict (torch tensors) -> flat dict of numpy arrays in this package's `.npz` layout."""
    to_np = lambda t: t.detach().cpu().float().numpy() if hasattr(t, "detach") else np.asarray(t, np.float32)
    blob = {}
    model = ckpt["model"]
    for k, v in model.items():
        if k.startswith(ALIAS) or k == "item_embedding.fused_embedding":   # aliases / inference-only fused table (model.py:120-137)
            continue
        blob["model/" + k] = to_np(v)
    shapes = {k: tuple(v.shape) for k, v in model.items()}
    if "optimizer" in ckpt:
        osd = ckpt["optimizer"]
        decay, nodecay = trainable_order(list(model.keys()), shapes)
        order = decay + nodecay
        groups = osd["param_groups"]
        assert [len(g["params"]) for g in groups] == [len(decay), len(nodecay)], "unexpected parameter groups"
        step = 0
        for idx, name in enumerate(order):
            st = osd["state"].get(idx)
            if st is None:
                continue
            assert tuple(st["exp_avg"].shape) == shapes[name], (name, st["exp_avg"].shape)
            blob["optimizer/exp_avg/" + name] = to_np(st["exp_avg"])
            blob["optimizer/exp_avg_sq/" + name] = to_np(st["exp_avg_sq"])
            step = int(float(st["step"]))
        blob["optimizer/step"] = np.array([step])
        blob["optimizer/lr"] = np.array([float(groups[0].get("initial_lr", groups[0]["lr"]))])
    if "scheduler" in ckpt:
        blob["scheduler/last_epoch"] = np.array([int(ckpt["scheduler"]["last_epoch"])])
    blob["config"] = np.frombuffer(json.dumps(ckpt.get("config", {}), default=float).encode(), np.uint8)
    blob["epoch"] = np.array([int(ckpt.get("epoch", -1))])
    blob["training_loss"] = np.array(ckpt.get("training_loss", []), np.float64)
    blob["test_loss"] = np.array(ckpt.get("test_loss", []), np.float64)
    return blob


def schedule_factor(lam, step):
    """learning-rate factor of the stored schedule parameters at `step` (WSD, transformer.py:310-328; a finetune
    checkpoint's constant schedule has no such parameters and gives 1)"""
    if "total_steps" not in lam:
        return 1.0
    s = max(0, min(int(step), lam["total_steps"]))
    if s <= lam["warmup_steps"]:
        return s / max(1, lam["warmup_steps"])
    plateau_end = lam["warmup_steps"] + lam["stable_steps"]
    if s <= plateau_end:
        return 1.0
    return 1.0 - (1.0 - lam["final_ratio"]) * ((s - plateau_end) / max(1, lam["decay_steps"]))


def to_reference(blob, scheduler_state=None, metadata=None):
    """`.npz`-layout dict -> reference checkpoint dict (torch tensors) that transformer.py:657-700 can resume from.

    * model: the reference's state-dict keys in its order, `watch_head.` aliases included.  The reference's strict
      `load_state_dict` also wants the frozen metadata table, which this package's checkpoints leave out (it is an input
      file, 4.9 GB at 200 K items): pass `metadata` = the (V, M) array of media_embeddings.h5 and it is inserted with its
      zero mask row (model.py:380-389); a blob that already holds the table keeps it.
    * optimizer: AdamW.state_dict() with the two parameter groups of create_optimizer; when the blob carries the
      schedule's parameters ("scheduler/lambda", written by train.checkpoint_model) the groups' `lr` is the SCHEDULED
      rate base * factor(last_epoch) and `initial_lr` the base rate, as torch's LambdaLR leaves them.
    * scheduler: a complete LambdaLR.state_dict() (`lr_lambdas` = the attribute dict of the reference's WSDScheduler per
      group, `base_lrs`, `_last_lr`, `last_epoch`, `_step_count`): torch's load_state_dict pops `lr_lambdas`.
      `scheduler_state` overrides it verbatim."""
    import torch
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32).copy())
    names = [k[len("model/"):] for k in blob if k.startswith("model/")]
    model = {n: T(blob["model/" + n]) for n in names}
    if metadata is not None and FROZEN not in model:
        meta = np.asarray(metadata, np.float32)
        table = np.zeros((meta.shape[0] + 1, meta.shape[1]), np.float32)
        table[:-1] = meta
        at = names.index("item_embedding.matchedid_embedding.embedding.weight") + 1
        names = names[:at] + [FROZEN] + names[at:]
        model[FROZEN] = torch.from_numpy(table)
    # state-dict order of the reference: aliases of the shared item embedding sit before the rating head
    ordered = {}
    for n in names:
        if n.startswith("rating_head.") and not any(k.startswith(ALIAS) for k in ordered):
            for m_ in names:
                if m_.startswith("item_embedding."):
                    ordered[ALIAS + m_] = model[m_]
        ordered[n] = model[n]
    if not any(k.startswith(ALIAS) for k in ordered):
        for m_ in names:
            if m_.startswith("item_embedding."):
                ordered[ALIAS + m_] = model[m_]
    ckpt = {"model": ordered}
    cfg = json.loads(bytes(np.asarray(blob["config"], np.uint8)).decode()) if "config" in blob else {}
    lam = json.loads(bytes(np.asarray(blob["scheduler/lambda"], np.uint8)).decode()) if "scheduler/lambda" in blob else None
    le = int(np.asarray(blob["scheduler/last_epoch"]).reshape(-1)[0]) if "scheduler/last_epoch" in blob else None
    factor = schedule_factor(lam, le) if (lam is not None and le is not None and scheduler_state is None) else None
    lr0 = float(np.asarray(blob.get("optimizer/lr", [cfg.get("learning_rate", 1e-4)])).reshape(-1)[0])
    n_groups = 2
    if "optimizer/step" in blob:
        shapes = {n: tuple(np.shape(blob["model/" + n])) for n in names if n != FROZEN}
        decay, nodecay = trainable_order([n for n in names if n != FROZEN], shapes)
        order = decay + nodecay
        step = float(np.asarray(blob["optimizer/step"]).reshape(-1)[0])
        state = {}
        for idx, n in enumerate(order):
            if "optimizer/exp_avg/" + n in blob:
                state[idx] = {"step": torch.tensor(step), "exp_avg": T(blob["optimizer/exp_avg/" + n]),
                              "exp_avg_sq": T(blob["optimizer/exp_avg_sq/" + n])}
        common = {"lr": lr0 if factor is None else lr0 * factor, "betas": (0.9, 0.95), "eps": 1e-8, "amsgrad": False,
                  "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": True,
                  "decoupled_weight_decay": True}
        if factor is not None:
            common["initial_lr"] = lr0
        ckpt["optimizer"] = {"state": state, "param_groups": [
            dict(weight_decay=0.1, **common, params=list(range(len(decay)))),
            dict(weight_decay=0.0, **common, params=list(range(len(decay), len(order))))]}
    if scheduler_state is not None:
        ckpt["scheduler"] = scheduler_state
    elif le is not None:
        f = 1.0 if factor is None else factor
        ckpt["scheduler"] = {"base_lrs": [lr0] * n_groups, "last_epoch": le, "_step_count": le + 1, "_is_initial": False,
                             "_get_lr_called_within_step": False, "_last_lr": [lr0 * f] * n_groups,
                             "lr_lambdas": [dict(lam) if lam is not None else None for _ in range(n_groups)]}
    ckpt["config"] = cfg
    ckpt["epoch"] = int(np.asarray(blob.get("epoch", [-1])).reshape(-1)[0])
    ckpt["training_loss"] = [float(x) for x in np.asarray(blob.get("training_loss", []))]
    ckpt["test_loss"] = [float(x) for x in np.asarray(blob.get("test_loss", []))]
    return ckpt


def load_metadata_table(path):
    """the (V, M) float32 `metadata` dataset of media_embeddings.h5 (transformer.jl:56-77), given as that file, the data
    directory that holds it, or a `.npy` array"""
    import os
    if path.endswith(".npy"):
        return np.load(path)
    if os.path.isdir(path):
        path = os.path.join(path, "media_embeddings.h5")
    from . import h5
    with h5.File(path) as f:
        return np.asarray(f["metadata"], np.float32)


def dedup_finetune_models(blobs):
    """Finetune/register.py:38-63 on `.npz`-layout dicts: the four finetuned checkpoints share one frozen trunk, so it is
    stored once (`base`) and every checkpoint keeps only its LoRA tensors.  Returns (base, [lora-only blobs]); raises like
    the reference when a checkpoint has no LoRA keys, no trunk keys, or a trunk that differs from the first one's."""
    base, out = None, []
    for i, blob in enumerate(blobs):
        trunk = {k: v for k, v in blob.items() if k.startswith("model/") and "lora_" not in k}
        lora = {k: v for k, v in blob.items() if k.startswith("model/") and "lora_" in k}
        assert lora, f"checkpoint {i}: no lora keys found"
        assert trunk, f"checkpoint {i}: no trunk keys found; is this a full checkpoint?"
        if base is None:
            base = trunk
        else:
            assert set(trunk) == set(base), f"checkpoint {i}: key mismatch"
            for k, v in trunk.items():
                assert np.array_equal(v, base[k]), f"checkpoint {i}: {k} differs from base"
        rest = {k: v for k, v in blob.items() if not k.startswith("model/")}
        rest.update(lora)
        out.append(rest)
    return base, out


def main(argv):
    if len(argv) not in (4, 5) or argv[1] not in ("pt2npz", "npz2pt"):
        print(__doc__)
        return 2
    import torch
    if argv[1] == "pt2npz":
        np.savez(argv[3], **from_reference(torch.load(argv[2], weights_only=False, map_location="cpu")))
    else:
        z = np.load(argv[2])
        meta = load_metadata_table(argv[4]) if len(argv) == 5 else None
        torch.save(to_reference({k: z[k] for k in z.files}, metadata=meta), argv[3])
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
