"""CPU: interchange with the reference's torch-pickle checkpoint.  The fixture
tests/golden/checkpoint_tiny.pt was written by the reference's own model / torch AdamW / LambdaLR after two training
steps (oracle/gen_golden.py); converting it to this package's layout and back must reproduce it exactly."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _load():
    torch = pytest.importorskip("torch")
    return torch, torch.load(os.path.join(GOLDEN, "checkpoint_tiny.pt"), weights_only=False, map_location="cpu")


def test_reference_checkpoint_round_trip():
    from recommendersystem_amd import checkpoint as ck
    torch, ref = _load()
    blob = ck.from_reference(ref)
    back = ck.to_reference(blob, scheduler_state=ref["scheduler"])
    assert list(back["model"].keys()) == list(ref["model"].keys())          # names AND state-dict order (aliases before the rating head)
    for k, v in ref["model"].items():
        assert torch.equal(back["model"][k], v), k
    assert [g["params"] for g in back["optimizer"]["param_groups"]] == [g["params"] for g in ref["optimizer"]["param_groups"]]
    for g0, g1 in zip(ref["optimizer"]["param_groups"], back["optimizer"]["param_groups"]):
        for key in ("lr", "betas", "eps", "weight_decay", "amsgrad", "maximize"):
            assert g0[key] == g1[key], key
    assert back["optimizer"]["state"].keys() == ref["optimizer"]["state"].keys()
    for i, st in ref["optimizer"]["state"].items():
        assert float(st["step"]) == float(back["optimizer"]["state"][i]["step"])
        assert torch.equal(st["exp_avg"], back["optimizer"]["state"][i]["exp_avg"])
        assert torch.equal(st["exp_avg_sq"], back["optimizer"]["state"][i]["exp_avg_sq"])
    assert back["scheduler"] == ref["scheduler"] and back["epoch"] == ref["epoch"]
    assert back["training_loss"] == ref["training_loss"] and back["test_loss"] == ref["test_loss"]


def test_blob_uses_the_state_dict_names_of_the_model():
    from oracle import synth
    from recommendersystem_amd import checkpoint as ck
    _, ref = _load()
    blob = ck.from_reference(ref)
    cfg = synth.make_config("tiny", mask_rate=0.25, mask_topk=6)
    names = list(synth.param_shapes(cfg).keys())
    assert [k[6:] for k in blob if k.startswith("model/")] == names
    for n in synth.trainable_names(cfg):
        assert blob["optimizer/exp_avg/" + n].shape == tuple(synth.param_shapes(cfg)[n])
    assert int(blob["optimizer/step"][0]) == 2 and int(blob["scheduler/last_epoch"][0]) == 2
    # weight-decay grouping of create_optimizer (transformer.py:285-298): matrices first, then vectors
    decay, nodecay = ck.trainable_order(names, {n: tuple(synth.param_shapes(cfg)[n]) for n in names})
    assert all(len(synth.param_shapes(cfg)[n]) >= 2 for n in decay) and all(len(synth.param_shapes(cfg)[n]) < 2 for n in nodecay)
    assert len(decay) == 22 and len(nodecay) == 11


def test_committed_conversion_is_current():
    """tests/golden/checkpoint_tiny_converted.npz (what the GPU resume test loads, so that it needs no torch) is exactly
    from_reference(checkpoint_tiny.pt)."""
    from recommendersystem_amd import checkpoint as ck
    _, ref = _load()
    blob = ck.from_reference(ref)
    z = np.load(os.path.join(GOLDEN, "checkpoint_tiny_converted.npz"))
    assert sorted(z.files) == sorted(blob.keys())
    for k in z.files:
        np.testing.assert_array_equal(z[k], blob[k], err_msg=k)


def test_dedup_finetune_models():
    """register.py:38-63: one shared trunk + per-checkpoint LoRA tensors; a trunk that differs is refused."""
    from recommendersystem_amd.checkpoint import dedup_finetune_models
    rng = np.random.default_rng(0)
    trunk = {"model/transformers.layers.0.attn.q_proj.weight": rng.standard_normal((4, 4)).astype(np.float32),
             "model/rating_head.0.bias": rng.standard_normal(4).astype(np.float32)}
    blobs = []
    for i in range(4):
        b = dict(trunk)
        b["model/transformers.layers.0.attn.q_proj_lora_A.weight"] = np.full((2, 4), i, np.float32)
        b["epoch"] = np.array([i])
        blobs.append(b)
    base, loras = dedup_finetune_models(blobs)
    assert sorted(base) == sorted(trunk) and len(loras) == 4
    for i, l in enumerate(loras):
        assert sorted(l) == ["epoch", "model/transformers.layers.0.attn.q_proj_lora_A.weight"] and l["epoch"][0] == i
    blobs[2]["model/rating_head.0.bias"] = blobs[2]["model/rating_head.0.bias"] + 1
    with pytest.raises(AssertionError):
        dedup_finetune_models(blobs)
    with pytest.raises(AssertionError):
        dedup_finetune_models([trunk])
