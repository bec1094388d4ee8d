"""Pins the numpy oracle (oracle/) to fixtures produced by RUNNING the reference's
transformer.model.py / transformer.py functions (oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import model_np, synth, train_np

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _case(name):
    z = np.load(os.path.join(GOLDEN, f"model_{name}.npz"))
    over = dict(tiny=dict(mask_rate=0.25, mask_topk=6), hd64=dict(mask_rate=0.2, mask_topk=16))[name]
    cfg = synth.make_config(name, **over)
    seed = int(z["meta/seed"][0]); rows = int(z["meta/rows"][0])
    P = synth.make_params(cfg, seed, "test")
    d = synth.make_batch(cfg, rows, seed + 1)
    u = z["meta/u"]
    r = np.float32(cfg["mask_rate"])
    wm = u < r
    rm = (u >= r) & (u < 2 * r)
    return z, cfg, P, d, wm, rm


def summarize(a):
    a = np.asarray(a, np.float64).reshape(-1)
    idx = np.linspace(0, a.size - 1, num=min(64, a.size)).astype(np.int64)
    return np.concatenate([[a.sum(), np.sqrt((a * a).sum()), np.abs(a).max()],
                           a[:32] if a.size >= 32 else np.pad(a, (0, 32 - a.size)),
                           a[idx] if idx.size == 64 else np.pad(a[idx], (0, 64 - idx.size))])


def close(a, b, rtol=2e-4, atol=None):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    atol = atol if atol is not None else rtol * scale
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", ["tiny", "hd64"])
def test_mask_tokens_bit_exact(name):
    z, cfg, P, d, wm, rm = _case(name)
    dm = model_np.mask_tokens(cfg, model_np.reshape_batch(cfg, d), wm, rm)
    for k in dm:
        ref = z["masked/" + k]
        assert dm[k].dtype == ref.dtype, k
        np.testing.assert_array_equal(dm[k], ref, err_msg=k)


@pytest.mark.parametrize("name", ["tiny", "hd64"])
def test_forward_and_grads(name):
    z, cfg, P, d, wm, rm = _case(name)
    model = model_np.OracleModel(cfg, P, np.float64)
    dm = model_np.mask_tokens(cfg, model_np.reshape_batch(cfg, d), wm, rm)
    y, ctx = model.embed(dm)
    if name == "tiny":
        close(ctx["e_a"], z["act/action_embedding"])
        close(ctx["e_i"], z["act/item_embedding"])
        close(y, z["act/trunk_out"])
    else:
        close(summarize(ctx["e_a"]), z["sum/action_embedding"])
        close(summarize(y), z["sum/trunk_out"])
    losses = model.forward(dm, False)
    close(losses, z["loss/train"], rtol=1e-5)
    ev = model.forward(dm, True)
    flat = []
    for x in ev:
        flat += list(x) if isinstance(x, list) else [x]
    close(flat, z["loss/eval"], rtol=1e-5)
    losses, G = model.forward(dm, False, True, z["meta/task_w"])
    names = synth.trainable_names(cfg)
    assert set(names) <= set(G)
    for n in names:
        if name == "tiny":
            close(G[n], z["grad/" + n], rtol=5e-4)
        else:
            close(summarize(G[n]), z["gsum/" + n], rtol=5e-4)
    gn = np.sqrt(sum((G[n] ** 2).sum() for n in names))
    close(gn, z["grad_norm"][0], rtol=1e-5)


@pytest.mark.parametrize("name", ["tiny", "hd64"])
def test_clip_adamw_three_steps(name):
    z, cfg, P, d, wm, rm = _case(name)
    names = synth.trainable_names(cfg)
    P = {k: np.asarray(v, np.float64) for k, v in P.items()}
    opt = train_np.AdamW(P, names, lr=float(z["opt/lr"][0]))
    for step in range(3):
        P, losses, norm = train_np.train_step(cfg, P, opt, d, wm, rm, z["meta/task_w"])
        close(losses, z["opt/losses"][step], rtol=2e-4)
        close(norm, z["opt/norms"][step], rtol=2e-4)
    for n in names:
        if name == "tiny":
            close(P[n], z["opt/param/" + n], rtol=2e-4)
        else:
            close(summarize(P[n]), z["opt/psum/" + n], rtol=2e-4)


def test_inference_rope_pos_and_candidates():
    z = np.load(os.path.join(GOLDEN, "infer_tiny.npz"))
    cfg = synth.make_config("tiny", mask_rate=0.25, mask_topk=6)
    P = synth.make_params(cfg, 31, "test")
    model = model_np.OracleModel(cfg, P, np.float64)
    d = {k[3:]: z[k] for k in z.files if k.startswith("in/")}
    close(model.inference(d, "retrieval"), z["out/retrieval"])
    close(model.inference(d, "ranking"), z["out/ranking"])


def test_finetune_lora_grads():
    z = np.load(os.path.join(GOLDEN, "finetune_tiny.npz"))
    cfg = synth.make_config("tiny", mask_rate=0.25, mask_topk=6, finetune=True, finetune_metric="rating")
    P = synth.make_params(cfg, 41, "test")
    model = model_np.OracleModel(cfg, P, np.float64)
    d = {k[3:]: z[k] for k in z.files if k.startswith("in/")}
    dm = model_np.mask_tokens(cfg, d)
    losses, G = model.forward(dm, False, True, [0.0, 1.0, 0.0, 0.5])
    close(losses, z["loss/train"], rtol=1e-5)
    gk = [k for k in z.files if k.startswith("grad/")]
    assert gk and all("lora_" in k for k in gk)
    for k in gk:
        close(G[k[5:]], z[k], rtol=5e-4)


def test_host_functions():
    z = np.load(os.path.join(GOLDEN, "host_fns.npz"))
    f = [train_np.wsd_factor(int(s), 2000, 50000) for s in z["wsd/steps"]]
    np.testing.assert_allclose(f, z["wsd/factors"], rtol=0, atol=1e-15)
    f2 = [train_np.wsd_factor(s, 10, 57) for s in range(60)]
    np.testing.assert_allclose(f2, z["wsd2/factors"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(train_np.make_task_weights(), z["task_w/pretrain"], rtol=1e-15)
    for med in (0, 1):
        for met in ("watch", "rating"):
            np.testing.assert_allclose(train_np.make_task_weights(med, met), z[f"task_w/{med}.{met}"], rtol=1e-15)
    out = [train_np.minimize_quadratic([1, 0, -1], list(y)) for y in z["minq/y"]]
    np.testing.assert_allclose(out, z["minq/out"], rtol=1e-12)
    perm = train_np.block_permutation_indices(z["shuffle/arr"], z["shuffle/block_perm"])
    np.testing.assert_array_equal(perm, z["shuffle/index_perm"])
    st = train_np.EarlyStopper(2, 0.001)
    for s, rec in zip(z["stopper/scores"], z["stopper/rec"]):
        st(float(s))
        assert [st.counter, float(st.early_stop), float(st.save_model)] == list(rec)


def test_known_answers_no_fixture():
    """doc-mask isolation, bidirectionality, token-mask isolation,
    and loss = ln(V_m) for a zero item table."""
    cfg = synth.make_config("tiny", mask_rate=0.25, mask_topk=6)
    P = synth.make_params(cfg, 5, "test")
    model = model_np.OracleModel(cfg, P)
    d = model_np.reshape_batch(cfg, synth.make_batch(cfg, 1, 9))
    S = cfg["max_sequence_length"]
    d["userid"][0, :] = np.where(np.arange(S) < S // 2, 1, 2)
    d["token_mask_ids"][:] = 0
    y0, _ = model.embed(d)
    d2 = {k: v.copy() for k, v in d.items()}
    d2["matchedid"][0, S - 1] = (d2["matchedid"][0, S - 1] + 1) % 30
    y1, _ = model.embed(d2)
    T = 2 * S
    assert np.abs(y1[0, :T // 2] - y0[0, :T // 2]).max() == 0          # other user untouched
    assert np.abs(y1[0, T // 2:T - 2] - y0[0, T // 2:T - 2]).max() > 0  # earlier tokens of same user change
    d3 = {k: v.copy() for k, v in d.items()}
    d3["token_mask_ids"][0, S - 1] = 1
    y2, _ = model.embed(d3)
    d4 = {k: v.copy() for k, v in d3.items()}
    d4["matchedid"][0, S - 1] = (d4["matchedid"][0, S - 1] + 1) % 30
    y3, _ = model.embed(d4)
    assert np.abs(y3[0, :T - 2] - y2[0, :T - 2]).max() == 0             # tmid=1 event invisible to others
    assert np.abs(y3[0, T - 2:] - y2[0, T - 2:]).max() > 0
    Pz = dict(P)
    for k in ("item_embedding.matchedid_embedding.embedding.weight", "item_embedding.projection_layer.weight",
              "item_embedding.projection_layer.bias"):
        Pz[k] = np.zeros_like(P[k])
    mz = model_np.OracleModel(cfg, Pz)
    wm, rm = synth.make_masks(cfg, 1, 3)
    wm[:] = True; rm[:] = False
    dm = model_np.mask_tokens(cfg, model_np.reshape_batch(cfg, synth.make_batch(cfg, 1, 9)), wm, rm)
    L = mz.forward(dm, False)
    if dm["0.watch.weight"].sum() > 0:
        assert abs(L[0] - np.log(30)) < 1e-9
    if dm["1.watch.weight"].sum() > 0:
        assert abs(L[2] - np.log(50)) < 1e-9
