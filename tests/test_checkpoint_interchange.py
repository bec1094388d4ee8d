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


def test_reference_can_resume_from_a_checkpoint_this_package_wrote(tmp_path):
    """train.checkpoint_model -> npz2pt -> what transformer.py:657-700 does with the file: strict load of the model keys
    (the reference's own state-dict key list, from the fixture it wrote, incl. the frozen table and the watch_head
    aliases), AdamW.load_state_dict on the two groups of create_optimizer, LambdaLR.load_state_dict (which pops
    `lr_lambdas`), and the next scheduler step continuing the curve at the scheduled rate."""
    from oracle import synth, train_np
    from recommendersystem_amd import checkpoint as ck
    from recommendersystem_amd import train as T
    torch, ref = _load()
    cfg = dict(ref["config"])
    P = synth.make_params(cfg, 5, "test")
    names = synth.trainable_names(cfg)
    rng = np.random.default_rng(1)

    class Model:
        def state_dict(self, include_frozen=True):
            return {k: v.astype(np.float32) for k, v in P.items() if include_frozen or k != ck.FROZEN}

    class Opt:
        def state_dict(self):
            return {"step": 7, "lr": 3e-3, "state": {n: {"exp_avg": rng.standard_normal(P[n].shape).astype(np.float32),
                                                         "exp_avg_sq": rng.random(P[n].shape).astype(np.float32)} for n in names}}

    sched = T.LambdaLR(T.WSDScheduler(warmup_steps=10, total_steps=40, decay_ratio=0.1, final_ratio=0.1))
    for _ in range(7):
        sched.step()
    T.checkpoint_model(str(tmp_path), Model(), Opt(), sched, cfg, 3, [1.0, 2.0, 3.0, 4.0], [1.5, 2.5, 3.5, 4.5], T.make_task_weights(), True)
    meta = np.asarray(P[ck.FROZEN][:-1], np.float32)
    np.save(tmp_path / "meta.npy", meta)
    assert ck.main(["", "npz2pt", str(tmp_path / "transformer.masked.npz"), str(tmp_path / "out.pt"), str(tmp_path / "meta.npy")]) == 0
    out = torch.load(tmp_path / "out.pt", weights_only=False, map_location="cpu")
    # (1) model.load_state_dict(checkpoint["model"]) is strict: same keys, same order, same shapes as the reference's own
    assert list(out["model"].keys()) == list(ref["model"].keys())
    for k, v in ref["model"].items():
        assert tuple(out["model"][k].shape) == tuple(v.shape), k
    np.testing.assert_array_equal(out["model"][ck.FROZEN].numpy()[:-1], meta)
    assert not out["model"][ck.FROZEN].numpy()[-1].any()                       # mask row (model.py:386)
    assert out["model"]["watch_head." + ck.FROZEN].data_ptr() == out["model"][ck.FROZEN].data_ptr() or \
        torch.equal(out["model"]["watch_head." + ck.FROZEN], out["model"][ck.FROZEN])
    # (2) optimizer.load_state_dict on create_optimizer's groups (transformer.py:285-298)
    shapes = {k: tuple(v.shape) for k, v in ref["model"].items()}
    decay, nodecay = ck.trainable_order(list(ref["model"].keys()), shapes)
    params = {n: torch.nn.Parameter(torch.zeros(shapes[n])) for n in decay + nodecay}
    opt = torch.optim.AdamW([{"params": [params[n] for n in decay], "weight_decay": 0.1},
                             {"params": [params[n] for n in nodecay], "weight_decay": 0.0}], lr=3e-3, betas=(0.9, 0.95))

    class RefSchedule:            # stands in for the reference's scheduler object: LambdaLR fills its __dict__ from `lr_lambdas`
        def __call__(self, step):
            return train_np.wsd_factor(step, self.warmup_steps, self.total_steps, self.decay_steps / self.total_steps, self.final_ratio)

    fresh = RefSchedule(); fresh.__dict__.update(warmup_steps=10, total_steps=40, decay_steps=4, final_ratio=0.1)
    lr_sched = torch.optim.lr_scheduler.LambdaLR(opt, fresh)
    opt.load_state_dict(out["optimizer"])
    lr_sched.load_state_dict(out["scheduler"])
    f7 = train_np.wsd_factor(7, 10, 40)
    assert [g["lr"] for g in opt.param_groups] == [3e-3 * f7] * 2 and [g["initial_lr"] for g in opt.param_groups] == [3e-3] * 2
    assert lr_sched.last_epoch == 7 and lr_sched.get_last_lr() == [3e-3 * f7] * 2
    assert vars(fresh)["stable_steps"] == 26 and vars(fresh)["decay_steps"] == 4
    st = opt.state_dict()["state"]
    assert len(st) == len(decay) + len(nodecay) and all(float(s["step"]) == 7.0 for s in st.values())
    opt.step(); lr_sched.step()
    assert lr_sched.get_last_lr() == [3e-3 * train_np.wsd_factor(8, 10, 40)] * 2
    assert out["epoch"] == 3 and out["config"] == cfg
    # and back: pt2npz of that file gives the moments under the parameter names again
    back = ck.from_reference(out)
    assert int(back["optimizer/step"][0]) == 7 and int(back["scheduler/last_epoch"][0]) == 7 and float(back["optimizer/lr"][0]) == 3e-3
