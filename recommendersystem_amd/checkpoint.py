"""Checkpoint interchange with the reference's `transformer.masked.pt`.

The reference writes a torch pickle (transformer.py:456-466):
    {"model": state_dict, "optimizer": AdamW.state_dict(), "scheduler": LambdaLR.state_dict(),
     "config", "epoch", "training_loss", "test_loss"}
(finetune checkpoints carry only model / config / epoch / losses, transformer.py:468-477) and resumes from it
(transformer.py:657-665, 690-700).  This package checkpoints to a flat `.npz` (train.checkpoint_model): the
reference's state-dict names under "model/", AdamW moments by parameter NAME, the scheduler's step counter.

    python -m recommendersystem_amd.checkpoint pt2npz transformer.masked.pt transformer.masked.npz
    python -m recommendersystem_amd.checkpoint npz2pt transformer.masked.npz transformer.masked.pt

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


def to_reference(blob, scheduler_state=None):
    """`.npz`-layout dict -> reference checkpoint dict (torch tensors).  The frozen metadata table is included when the
    blob has it (the reference's strict load wants it); `scheduler_state` (a LambdaLR.state_dict()) may be supplied,
    otherwise only `last_epoch` / `_step_count` are filled in."""
    import torch
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32).copy())
    names = [k[len("model/"):] for k in blob if k.startswith("model/")]
    model = {}
    for n in names:
        model[n] = T(blob["model/" + n])
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
    if "optimizer/step" in blob:
        shapes = {n: tuple(np.shape(blob["model/" + n])) for n in names}
        decay, nodecay = trainable_order(names, shapes)
        order = decay + nodecay
        step = float(np.asarray(blob["optimizer/step"]).reshape(-1)[0])
        lr = float(np.asarray(blob.get("optimizer/lr", [cfg.get("learning_rate", 1e-4)])).reshape(-1)[0])
        state = {}
        for idx, n in enumerate(order):
            if "optimizer/exp_avg/" + n in blob:
                state[idx] = {"step": torch.tensor(step), "exp_avg": T(blob["optimizer/exp_avg/" + n]),
                              "exp_avg_sq": T(blob["optimizer/exp_avg_sq/" + n])}
        common = {"lr": lr, "betas": (0.9, 0.95), "eps": 1e-8, "amsgrad": False, "maximize": False, "foreach": None,
                  "capturable": False, "differentiable": False, "fused": True, "decoupled_weight_decay": True}
        ckpt["optimizer"] = {"state": state, "param_groups": [
            dict(weight_decay=0.1, **common, params=list(range(len(decay)))),
            dict(weight_decay=0.0, **common, params=list(range(len(decay), len(order))))]}
    if scheduler_state is not None:
        ckpt["scheduler"] = scheduler_state
    elif "scheduler/last_epoch" in blob:
        le = int(np.asarray(blob["scheduler/last_epoch"]).reshape(-1)[0])
        ckpt["scheduler"] = {"last_epoch": le, "_step_count": le + 1}
    ckpt["config"] = cfg
    ckpt["epoch"] = int(np.asarray(blob.get("epoch", [-1])).reshape(-1)[0])
    ckpt["training_loss"] = [float(x) for x in np.asarray(blob.get("training_loss", []))]
    ckpt["test_loss"] = [float(x) for x in np.asarray(blob.get("test_loss", []))]
    return ckpt


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
    if len(argv) != 4 or argv[1] not in ("pt2npz", "npz2pt"):
        print(__doc__)
        return 2
    import torch
    if argv[1] == "pt2npz":
        np.savez(argv[3], **from_reference(torch.load(argv[2], weights_only=False, map_location="cpu")))
    else:
        z = np.load(argv[2])
        torch.save(to_reference({k: z[k] for k in z.files}), argv[3])
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
