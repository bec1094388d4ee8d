"""GPU: the integer / index paths of the HIP step read back through rsys_debug_get and compared BIT-EXACTLY with the
reference's own outputs (tests/golden/model_*.npz: masked/*, written by the reference's mask_tokens, model.py:417-462)
and with the oracle's position selection (model.py:501-513 made deterministic, oracle/model_np.select_positions)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TASK_W = [0.05, 0.2, 0.3, 0.25]
TASKS = [(m, k) for m in (0, 1) for k in ("watch", "rating")]

CASES = [
    ("tiny", dict(mask_rate=0.25, mask_topk=6), 3, 11),
    ("hd64", dict(mask_rate=0.2, mask_topk=16), 2, 23),
]


def _same_bits(a, b):
    a = np.ascontiguousarray(a).reshape(-1); b = np.ascontiguousarray(b).reshape(-1)
    assert a.dtype.itemsize == b.dtype.itemsize == 4, (a.dtype, b.dtype)
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _check_masked(model, rows, masked, what):
    """masked: dict name -> (rows, S) arrays as mask_tokens returns them"""
    for key, name, dt in (("masked.token_mask_ids", "token_mask_ids", np.int32), ("masked.matchedid", "matchedid", np.int32),
                          ("masked.status", "status", np.int32), ("masked.rating", "rating", np.float32),
                          ("masked.progress", "progress", np.float32)):
        assert _same_bits(model.debug_get(key, rows), np.asarray(masked[name]).astype(dt)), (what, key)
    for m, metric in TASKS:
        for field, dt in (("label", np.float32), ("weight", np.float32), ("position", np.int32)):
            got = model.debug_get(f"masked.{m}.{metric}.{field}", rows)
            assert _same_bits(got, np.asarray(masked[f"{m}.{metric}.{field}"]).astype(dt)), (what, m, metric, field)


def _check_selection_and_gather(model, cfg, rows, masked, what):
    from oracle import model_np
    S, D = cfg["max_sequence_length"], cfg["embed_dim"]
    V = cfg["vocab_sizes"]["0_matchedid"] + cfg["vocab_sizes"]["1_matchedid"]
    KB = cfg["mask_topk"] * rows
    npos = model.debug_get("npos", rows)
    for ti, (m, metric) in enumerate(TASKS):
        w = np.asarray(masked[f"{m}.{metric}.weight"]).reshape(-1)
        want = model_np.select_positions(w, KB).astype(np.int32)
        got = model.debug_get(f"idx.{ti}", rows)
        assert np.array_equal(got, want), (what, ti, got[:8], want[:8])
        assert int(npos[ti]) == min(int((w > 0).sum()), KB), (what, ti)
    # compact top of the trunk (training passes): the sorted union of the tasks' live tokens (2 i + metric) and its inverse map
    if int(model.debug_get("top.cap", rows)[0]) > 0:
        toks = []
        for ti, (m, metric) in enumerate(TASKS):
            got = model.debug_get(f"idx.{ti}", rows)[:int(npos[ti])]
            toks.append(2 * got.astype(np.int64) + (ti & 1))
        want = np.unique(np.concatenate(toks)).astype(np.int32)
        n_sel = int(model.debug_get("top.n", rows)[0])
        assert n_sel == want.size, (what, n_sel, want.size)
        assert np.array_equal(model.debug_get("top.sel", rows)[:n_sel], want), what
        slot = np.full(2 * rows * S, -1, np.int32); slot[want] = np.arange(want.size, dtype=np.int32)
        assert np.array_equal(model.debug_get("top.slot", rows), slot), what
    # interleave (model.py:403-415, 468-469): per-token userid / token_mask_ids, every event twice
    uid = np.asarray(masked["userid"]).reshape(-1).astype(np.int32)
    tm = np.asarray(masked["token_mask_ids"]).reshape(-1).astype(np.int32)
    assert np.array_equal(model.debug_get("tokens.userid", rows), np.repeat(uid, 2)), what
    assert np.array_equal(model.debug_get("tokens.token_mask_ids", rows), np.repeat(tm, 2)), what
    # item gather (model.py:23-24,139-145): even token rows are exactly the fused-table rows of the remapped ids
    ids = np.asarray(masked["matchedid"]).reshape(-1)
    ids = np.where(ids == -1, V, ids)
    x0 = model.debug_get("embed.x0", rows)
    F = model.debug_get("table.fused", rows)
    assert _same_bits(x0[0::2], F[ids]), what


@pytest.mark.parametrize("name,over,rows,seed", CASES)
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_mask_tokens_selection_gather_bit_exact_vs_reference_fixture(name, over, rows, seed, dtype):
    import recommendersystem_amd as ra
    from oracle import model_np, synth
    cfg = synth.make_config(name, **over)
    P = synth.make_params(cfg, seed, "test")
    d = synth.make_batch(cfg, rows, seed + 1)
    z = np.load(os.path.join(GOLDEN, f"model_{name}.npz"))
    u = z["meta/u"]; r = np.float32(cfg["mask_rate"])
    wm = u < r; rm = (u >= r) & (u < 2 * r)
    model = ra.RecommenderModel(cfg, dtype=dtype, max_rows=rows)
    model.load_state_dict(P)
    model.set_loss_weights(TASK_W, 1)
    model(d, False, masks=(wm, rm))
    golden = {k[len("masked/"):]: z[k] for k in z.files if k.startswith("masked/")}       # the reference's own output
    _check_masked(model, rows, golden, "golden")
    dm = model_np.mask_tokens(cfg, model_np.reshape_batch(cfg, d), wm, rm)
    _check_masked(model, rows, dm, "oracle")
    _check_selection_and_gather(model, cfg, rows, golden, name)
    model.close()


def test_position_selection_overflow_lowest_flat_index_first():
    """#(w > 0) > mask_topk * rows (the reference only asserts mask_topk > mask_rate * S, train.py:561, i.e. on average):
    the build's rule is "positive weights in ascending flat index"; npos saturates at mask_topk * rows; losses and
    gradients equal the oracle evaluated on the same positions."""
    import recommendersystem_amd as ra
    from oracle import model_np, synth
    cfg = synth.make_config("hd64", mask_rate=0.45, mask_topk=4)
    rows, seed = 3, 41
    P = synth.make_params(cfg, seed, "test")
    d = synth.make_batch(cfg, rows, seed + 1)
    wm, rm = synth.make_masks(cfg, rows, seed + 2)
    dm = model_np.mask_tokens(cfg, model_np.reshape_batch(cfg, d), wm, rm)
    KB = cfg["mask_topk"] * rows
    over = [int((np.asarray(dm[f"{m}.{k}.weight"]) > 0).sum()) for m, k in TASKS]
    assert max(over) > KB, over                                  # the case under test really overflows
    model = ra.RecommenderModel(cfg, dtype="fp32", max_rows=rows)
    model.load_state_dict(P)
    model.set_loss_weights(TASK_W, 1)
    losses = model(d, False, masks=(wm, rm))
    _check_masked(model, rows, dm, "overflow")
    _check_selection_and_gather(model, cfg, rows, dm, "overflow")
    npos = model.debug_get("npos", rows)
    for ti, (m, k) in enumerate(TASKS):
        w = np.asarray(dm[f"{m}.{k}.weight"]).reshape(-1)
        if over[ti] > KB:
            assert int(npos[ti]) == KB
            assert np.array_equal(model.debug_get(f"idx.{ti}", rows), np.flatnonzero(w > 0)[:KB].astype(np.int32))
    ref = model_np.OracleModel(cfg, P, np.float64)
    l_ref, G_ref = ref.forward(dm, False, True, TASK_W)
    for a, b in zip(losses, l_ref):
        assert abs(a - b) <= 1e-4 * max(abs(b), 1.0), (losses, l_ref)
    for n in ("item_embedding.matchedid_embedding.embedding.weight", "rating_head.0.weight", "transformers.layers.0.mlp.w2.weight"):
        g = model.grad(n)
        assert np.abs(g - G_ref[n]).max() <= 5e-4 * max(np.abs(G_ref[n]).max(), 1e-12), n
    model.close()


def test_device_drawn_masks_follow_mask_tokens_exactly():
    """Without explicit masks the kernel draws u ~ U[0,1) per interaction from Philox (model.py:437-440).  Recover the two
    masks from its outputs, then every masked array must equal mask_tokens(those masks) bit for bit; the masks are
    disjoint, hit the requested rate, differ between steps and repeat for the same (seed, step)."""
    import recommendersystem_amd as ra
    from oracle import model_np, synth
    cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16)
    rows = 6
    d = synth.make_batch(cfg, rows, 5)
    for k in ("status",):                                         # status -1 never occurs in shards (import_list.jl:9-19)
        assert (np.asarray(d[k]) >= 0).all()
    model = ra.RecommenderModel(cfg, dtype="bf16", max_rows=rows)
    model.init_weights(3)
    model.set_loss_weights(TASK_W, 1)
    model.upload(d)
    drawn = []
    for step in (0, 1, 0):
        model.zero_grad()
        model.forward_resident(False, step=step)
        model.losses(False)
        any_mask = model.debug_get("masked.status", rows) == -1
        wm = (model.debug_get("masked.matchedid", rows) == -1) & (np.asarray(d["matchedid"]).reshape(-1) != -1)
        rm = any_mask & ~wm
        assert not (wm & ~any_mask).any()
        shape = (rows, cfg["max_sequence_length"])
        dm = model_np.mask_tokens(cfg, model_np.reshape_batch(cfg, d), wm.reshape(shape), rm.reshape(shape))
        _check_masked(model, rows, dm, f"step{step}")
        _check_selection_and_gather(model, cfg, rows, dm, f"step{step}")
        n = wm.size
        assert abs(wm.mean() - 0.2) < 4 * np.sqrt(0.16 / n) and abs(rm.mean() - 0.2) < 4 * np.sqrt(0.16 / n)
        drawn.append((wm, rm))
    assert np.array_equal(drawn[0][0], drawn[2][0]) and np.array_equal(drawn[0][1], drawn[2][1])
    assert not np.array_equal(drawn[0][0], drawn[1][0])
    model.close()


def test_finetune_masks_come_from_the_metric_weights():
    """model.py:418-435: in finetune mode the masked positions are those with a positive weight of the chosen metric."""
    import recommendersystem_amd as ra
    from oracle import model_np, synth
    for metric in ("watch", "rating"):
        cfg = synth.make_config("tiny", mask_rate=0.25, mask_topk=6)
        cfg["finetune"] = True; cfg["finetune_metric"] = metric; cfg["lora_dropout"] = 0.0
        rows = 3
        d = synth.make_batch(cfg, rows, 19)
        rng = np.random.default_rng(2)
        for m in (0, 1):                                          # finetune shards: few targets, fractional weights
            for k in ("watch", "rating"):
                w = np.asarray(d[f"{m}.{k}.weight"])
                d[f"{m}.{k}.weight"] = (w * (rng.random(w.shape) < 0.2) * rng.random(w.shape)).astype(np.float32)
        model = ra.RecommenderModel(cfg, dtype="fp32", max_rows=rows)
        model.init_weights(3)
        model.set_loss_weights(TASK_W, 1)
        model(d, False)
        dm = model_np.mask_tokens(cfg, model_np.reshape_batch(cfg, d))
        _check_masked(model, rows, dm, metric)
        model.close()


def test_rejected_batch_leaves_the_resident_one_untouched():
    """rsys_batch_upload validates every index array on the host before it copies anything: after a rejected upload the
    previous batch is still resident and a forward over it gives the same result as before."""
    import recommendersystem_amd as ra
    from oracle import synth
    from recommendersystem_amd._lib import RsysError
    cfg = synth.make_config("tiny", mask_rate=0.25, mask_topk=6)
    rows = 3
    d = synth.make_batch(cfg, rows, 12)
    wm, rm = synth.make_masks(cfg, rows, 13)
    model = ra.RecommenderModel(cfg, dtype="fp32", max_rows=rows)
    model.init_weights(3)
    model.set_loss_weights(TASK_W, 1)
    l0 = model(d, True, masks=(wm, rm))
    before = model.debug_get("masked.matchedid", rows).copy()
    bad = {k: np.array(v, copy=True) for k, v in d.items()}
    bad["matchedid"][5] = cfg["vocab_sizes"]["0_matchedid"] + cfg["vocab_sizes"]["1_matchedid"] + 7
    with pytest.raises(RsysError, match="matchedid out of range"):
        model.upload(bad, (wm, rm))
    model.forward_resident(True, step=0)
    l1 = model.losses(True)                                      # (loss sums use float atomics: equal to rounding, not bitwise)
    flat = lambda ls: [x for e in ls for x in (e if isinstance(e, list) else [e])]
    assert np.allclose(flat(l1), flat(l0), rtol=1e-5, atol=0)
    assert np.array_equal(model.debug_get("masked.matchedid", rows), before)
    model.close()
