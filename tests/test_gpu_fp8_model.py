"""GPU: the fp8 trunk (dtype="fp8": the reference's pretraining arithmetic, torchao "tensorwise" float8 linears under `transformers.`,
transformer.py:671-676) against the numpy oracle run with the same recipe (oracle/model_np.py operand_round="fp8", oracle/fp8.py).
The oracle's rounding and linear products are pinned to torch's float8 casts and torch._scaled_mm (tests/test_fp8_oracle.py); torchao
itself is absent from the image, so its amax -> scale formula is restated.  What these tests pin is that the HIP path computes the
recipe as restated -- and that it is the fp8 arithmetic, not bf16 (the two oracles differ by far more than HIP differs from its own)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TASK_W = [0.05, 0.2, 0.3, 0.25]


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _grad_errors(model, G_ref, names):
    out = {}
    for n in names:
        g = model.grad(n)
        scale = max(np.abs(G_ref[n]).max(), 1e-3 * np.sqrt((G_ref[n] ** 2).mean()) + 1e-12)
        out[n] = float(np.abs(g - G_ref[n]).max() / max(scale, 1e-6))
    return out


def _rms(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-30)))


def _split_ab(ab, Ip):
    """[16 a | 16 b] column blocks -> (a, b)"""
    r = ab.reshape(ab.shape[0], Ip // 16, 2, 16)
    return r[:, :, 0, :].reshape(ab.shape[0], Ip), r[:, :, 1, :].reshape(ab.shape[0], Ip)


@pytest.mark.parametrize("hidden", [384, 352])   # 352: not a multiple of the 128-element K tile -- the hidden width is zero-padded inside
def test_every_fp8_product_of_the_trunk_stage_by_stage(monkeypatch, hidden):
    """Each of the eight fp8 products of a layer, fed with the HIP path's OWN stored operands, against the oracle's linear on those
    operands: no error is carried from stage to stage, so the bounds are the output rounding (bf16: relative RMS ~ 1e-3) or the
    accumulation (fp32 outputs: 1e-4) -- a wrong scale slot, weight copy, segment boundary or K order would be orders larger."""
    monkeypatch.setenv("RSYS_F8_DEBUG_KEEP", "1")
    import recommendersystem_amd as ra
    from oracle import model_np, synth
    cfg = synth.make_config("f8t", mask_rate=0.2, mask_topk=16, intermediate_dim=hidden)
    rows, seed = 3, 31
    P = synth.make_params(cfg, seed, "test")
    d = synth.make_batch(cfg, rows, seed + 1)
    wm, rm = synth.make_masks(cfg, rows, seed + 2)
    ref = model_np.OracleModel(cfg, P, np.float64, operand_round="fp8")
    Q = ref.q
    I = hidden
    model = ra.RecommenderModel(cfg, dtype="fp8", max_rows=rows)
    model.load_state_dict(P)
    model.set_loss_weights(TASK_W, 1)
    model(d, False, masks=(wm, rm))
    L, D, H, KV = cfg["num_layers"], cfg["embed_dim"], cfg["num_heads"], cfg["num_kv_heads"]
    hd, Ip, B, T = D // H, (cfg["intermediate_dim"] + 127) // 128 * 128, rows, 2 * cfg["max_sequence_length"]
    cos, sin = ref.cos[:T].astype(np.float64), ref.sin[:T].astype(np.float64)
    get = lambda k: model.debug_get(k, rows).astype(np.float64)
    worst_bf, worst_f32 = ("", 0.0), ("", 0.0)

    def chk(tag, got, want, f32=False):
        nonlocal worst_bf, worst_f32
        e = _rms(got, want)
        if f32:
            worst_f32 = max(worst_f32, (tag, e), key=lambda t: t[1])
        else:
            worst_bf = max(worst_bf, (tag, e), key=lambda t: t[1])
    for l in range(L):
        p = f"transformers.layers.{l}."
        xn = get(f"act.{l}.xn")
        q = Q(model_np.apply_rope(ref.lin(xn, p + "attn.q_proj.weight").reshape(B, T, H, hd), cos, sin)).reshape(B * T, -1)
        k = Q(model_np.apply_rope(ref.lin(xn, p + "attn.k_proj.weight").reshape(B, T, KV, hd), cos, sin)).reshape(B * T, -1)
        v = Q(ref.lin(xn, p + "attn.v_proj.weight"))
        chk(f"qkv_fwd[{l}]", get(f"act.{l}.qkv"), np.concatenate([q, k, v], 1))
        h = get(f"act.{l}.x") + ref.lin(get(f"act.{l}.O"), p + "attn.output_proj.weight")
        chk(f"o_fwd[{l}]", get(f"act.{l}.h"), h, f32=True)
        hn = get(f"act.{l}.hn")
        a = ref.lin(hn, p + "mlp.w1.weight"); b = ref.lin(hn, p + "mlp.w3.weight")
        a_hip, b_hip = (t[:, :I] for t in _split_ab(get(f"act.{l}.ab"), Ip))
        g_hip = get(f"act.{l}.g")
        assert not g_hip[:, I:].any()                                  # (padding columns stay zero)
        g_hip = g_hip[:, :I]
        chk(f"w1_fwd[{l}]", a_hip, Q(a)); chk(f"w3_fwd[{l}]", b_hip, Q(b))
        chk(f"swiglu[{l}]", g_hip, Q(a / (1.0 + np.exp(-a)) * b))
        if l + 1 < L:
            chk(f"w2_fwd[{l}]", get(f"act.{l + 1}.x"), get(f"act.{l}.h") + ref.lin(g_hip, p + "mlp.w2.weight"), f32=True)
        # backward: dY operands kept by the deferred weight gradients + the kept dx outputs
        gg = ref.lin_dx(get(f"dw.{l}.gxt"), p + "mlp.w2.weight")
        sig = 1.0 / (1.0 + np.exp(-a_hip))
        da_hip, db_hip = (t[:, :I] for t in _split_ab(get(f"dw.{l}.dab"), Ip))
        chk(f"w2_dx.da[{l}]", da_hip, Q(gg * b_hip * (sig * (1.0 + a_hip * (1.0 - sig)))))
        chk(f"w2_dx.db[{l}]", db_hip, Q(gg * a_hip * sig))
        chk(f"w13_dx[{l}]", get(f"f8keep.{l}.0"), Q(ref.lin_dx(da_hip, p + "mlp.w1.weight") + ref.lin_dx(db_hip, p + "mlp.w3.weight")))
        chk(f"o_dx[{l}]", get(f"f8keep.{l}.1"), Q(ref.lin_dx(get(f"dw.{l}.dht"), p + "attn.output_proj.weight")))
        dqkv = get(f"dw.{l}.dqkv")
        nq, nk = H * hd, KV * hd
        chk(f"qkv_dx[{l}]", get(f"f8keep.{l}.2"), Q(ref.lin_dx(dqkv[:, :nq], p + "attn.q_proj.weight") + ref.lin_dx(dqkv[:, nq:nq + nk], p + "attn.k_proj.weight")
                                                     + ref.lin_dx(dqkv[:, nq + nk:], p + "attn.v_proj.weight")))
        # the seven weight gradients (fp32 sums) from the operands the two passes stored
        grad = lambda n: model.grad(p + n).astype(np.float64)
        chk(f"w2_dw[{l}]", grad("mlp.w2.weight"), ref.lin_dw(get(f"dw.{l}.gxt"), g_hip), f32=True)
        chk(f"w1_dw[{l}]", grad("mlp.w1.weight"), ref.lin_dw(da_hip, hn), f32=True)
        chk(f"w3_dw[{l}]", grad("mlp.w3.weight"), ref.lin_dw(db_hip, hn), f32=True)
        chk(f"o_dw[{l}]", grad("attn.output_proj.weight"), ref.lin_dw(get(f"dw.{l}.dht"), get(f"act.{l}.O")), f32=True)
        chk(f"q_dw[{l}]", grad("attn.q_proj.weight"), ref.lin_dw(dqkv[:, :nq], xn), f32=True)
        chk(f"k_dw[{l}]", grad("attn.k_proj.weight"), ref.lin_dw(dqkv[:, nq:nq + nk], xn), f32=True)
        chk(f"v_dw[{l}]", grad("attn.v_proj.weight"), ref.lin_dw(dqkv[:, nq + nk:], xn), f32=True)
    # the amax every producer left beside its output (RMSNorm fwd / bwd, attention fwd / bwd, the two SwiGLU epilogues) is exactly
    # the amax of the tensor it stored
    am = model.debug_get("f8.aamax", rows).max(1)               # [layer][slot]
    nq, nk = H * hd, KV * hd
    for l in range(L):
        da_hip, db_hip = _split_ab(get(f"dw.{l}.dab"), Ip)
        dqkv = get(f"dw.{l}.dqkv")
        want = {0: get(f"act.{l}.xn"), 1: get(f"act.{l}.O"), 2: get(f"act.{l}.hn"), 3: get(f"act.{l}.g"), 4: get(f"dw.{l}.gxt"), 5: da_hip, 6: db_hip,
                7: get(f"dw.{l}.dht"), 8: dqkv[:, :nq], 9: dqkv[:, nq:nq + nk], 10: dqkv[:, nq + nk:]}
        for slot, t in want.items():
            assert am[l, slot] == np.float32(np.abs(t).max()), (l, slot, am[l, slot], np.abs(t).max())
    print("stage-wise fp8 products: worst bf16-output stage", worst_bf, "worst fp32-output stage", worst_f32)
    assert worst_bf[1] < 3e-3, worst_bf        # one bf16 rounding of the output: ~1.1e-3 relative RMS
    assert worst_f32[1] < 1e-4, worst_f32
    model.close()


def test_fp8_trunk_vs_fp8_oracle():
    """End to end.  Every quantisation step turns a bf16-ulp disagreement between the two implementations (summation order) into a
    whole fp8 step on the few elements that sit at a rounding boundary, so the agreement after two layers and a backward pass is
    percent-level by nature; the stage-wise test above is the sharp one.  Here: the result is finite, close to the fp8 oracle in
    RMS, and clearly closer to it than the bf16 arithmetic is."""
    import recommendersystem_amd as ra
    from oracle import model_np, synth
    cfg = synth.make_config("f8t", mask_rate=0.2, mask_topk=16)
    rows, seed = 3, 31
    P = synth.make_params(cfg, seed, "test")
    d = synth.make_batch(cfg, rows, seed + 1)
    wm, rm = synth.make_masks(cfg, rows, seed + 2)
    dm = model_np.mask_tokens(cfg, model_np.reshape_batch(cfg, d), wm, rm)
    res = {}
    for mode in ("fp8", "bf16"):
        ref = model_np.OracleModel(cfg, P, np.float64, operand_round=mode)
        y, _ = ref.embed(dm)
        l, G = ref.forward(dm, False, True, TASK_W)
        res[mode] = (y, l, G)
    y_ref, l_ref, G_ref = res["fp8"]
    names = synth.trainable_names(cfg)
    model = ra.RecommenderModel(cfg, dtype="fp8", max_rows=rows)
    model.load_state_dict(P)
    model.set_loss_weights(TASK_W, 1)
    losses = model(d, False, masks=(wm, rm))
    y = model.trunk_output(rows)
    e_y, gap_y = _rms(y, y_ref), _rms(res["bf16"][0], y_ref)
    e_l = max(abs(a - b) / max(abs(b), 1.0) for a, b in zip(losses, l_ref))
    ratios = {n: (_rms(model.grad(n), G_ref[n]), _rms(res["bf16"][2][n], G_ref[n])) for n in names}
    worst = max(ratios.items(), key=lambda kv: kv[1][0] / max(kv[1][1], 1e-9))
    print(f"fp8 end to end: trunk rms {e_y:.2e} (fp8 | bf16 oracles apart {gap_y:.2e}) losses {e_l:.2e}; gradient furthest from its oracle relative to the gap {worst}")
    assert np.isfinite(y).all() and all(np.isfinite(model.grad(n)).all() for n in names)
    assert e_y < 0.6 * gap_y and e_y < 6e-2, (e_y, gap_y)
    assert e_l < 5e-2, (losses, l_ref)
    for n, (e, gap) in ratios.items():
        assert e < 0.75 * gap + 2e-2, (n, e, gap)
    model.close()


def test_fp8_trunk_with_bf16_weight_gradients(monkeypatch):
    """RSYS_F8_DW=0 (Switches::f8_dw): the fp8 trunk takes its weight gradients from the bf16 operands instead of the transposed fp8
    copies (the path deterministic mode uses).  The forward pass is untouched -- the same losses -- and every gradient stays within
    the fp8 quantisation of the default arm."""
    import recommendersystem_amd as ra
    from oracle import synth
    cfg = synth.make_config("f8t", mask_rate=0.2, mask_topk=16)
    rows, seed = 3, 41
    P = synth.make_params(cfg, seed, "test")
    d = synth.make_batch(cfg, rows, seed + 1); mk = synth.make_masks(cfg, rows, seed + 2)
    names = synth.trainable_names(cfg)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("RSYS_F8_DW", flag)
        model = ra.RecommenderModel(cfg, dtype="fp8", max_rows=rows)
        model.load_state_dict(P)
        model.set_loss_weights(TASK_W, 1)
        losses = model(d, False, masks=mk)
        res[flag] = (np.array(losses), {n: model.grad(n).copy() for n in names})
        model.close()
    assert np.allclose(res["0"][0], res["1"][0], rtol=1e-6, atol=0), (res["0"][0], res["1"][0])   # (the loss sums use float atomics)
    worst = max((_rms(res["0"][1][n], res["1"][1][n]), n) for n in names)
    assert worst[0] < 8e-2, worst
    assert any(not np.array_equal(res["0"][1][n], res["1"][1][n]) for n in names)      # the switch did select another path


def test_fp8_is_refused_where_the_reference_does_not_use_it():
    import recommendersystem_amd as ra
    from oracle import synth
    with pytest.raises(ra.RsysError):      # finetuning keeps bf16 linears (transformer.py:671: `if not config["finetune"]`)
        ra.RecommenderModel(synth.make_config("f8t", finetune=True, finetune_metric="rating"), dtype="fp8", max_rows=2)
    with pytest.raises(ra.RsysError):      # shapes the 128-element K tiles cannot take
        ra.RecommenderModel(synth.make_config("hd64"), dtype="fp8", max_rows=2)


def test_fp8_training_tracks_bf16():
    """48 optimizer steps from the same initial state on the same batches: the smoothed fp8 loss curve stays inside a band around the
    bf16 one (single steps of two runs differ by more than the two arithmetics do: float atomics, a small model, a high rate)"""
    import recommendersystem_amd as ra
    from oracle import synth
    cfg = synth.make_config("f8t", mask_rate=0.2, mask_topk=16, learning_rate=5e-4)
    rows = 4
    P = synth.make_params(cfg, 5, "test")
    batches = [synth.make_batch(cfg, rows, 100 + i) for i in range(8)]
    masks = [synth.make_masks(cfg, rows, 200 + i) for i in range(8)]
    curves = {}
    for dtype in ("bf16", "fp8"):
        model = ra.RecommenderModel(cfg, dtype=dtype, max_rows=rows)
        model.load_state_dict(P)
        opt = ra.create_optimizer(model, cfg)
        model.set_loss_weights(TASK_W, 1)
        cur = []
        for step in range(48):
            losses = model(batches[step % 8], False, masks=masks[step % 8])
            opt.step(clip_max_norm=1.0)
            cur.append(sum(w * l for w, l in zip(TASK_W, losses)))
        curves[dtype] = np.array(cur)
        model.close()
    a, b = curves["bf16"], curves["fp8"]
    sm = lambda x: np.convolve(x, np.ones(8) / 8.0, mode="valid")      # one pass over the 8 batches
    sa, sb = sm(a), sm(b)
    print("weighted loss (mean of 8 steps), start / middle / end: bf16", sa[[0, 20, -1]], "fp8", sb[[0, 20, -1]])
    assert np.isfinite(b).all()
    assert sb[-1] < 0.8 * sb[0]                                          # it learns
    assert np.abs(sb - sa).max() < 0.15 * sa[0], (np.abs(sb - sa).max(), sa[0])


def test_fp8_deterministic_mode_is_bitwise_reproducible():
    """amax is a maximum (order-free) and every other reduction takes the deterministic path of the bf16 mode; the weight gradients
    then come from the bf16 operands through the ordered split-K slabs (the fp8 split-K form adds with float atomics)"""
    import recommendersystem_amd as ra
    from oracle import synth
    cfg = synth.make_config("f8t", mask_rate=0.2, mask_topk=16, deterministic=True)
    rows = 3
    P = synth.make_params(cfg, 9, "test")
    batches = [synth.make_batch(cfg, rows, 60 + i) for i in range(2)]
    masks = [synth.make_masks(cfg, rows, 70 + i) for i in range(2)]
    names = synth.trainable_names(cfg)

    def run():
        model = ra.RecommenderModel(cfg, dtype="fp8", max_rows=rows)
        model.load_state_dict(P)
        opt = ra.create_optimizer(model, dict(cfg, learning_rate=1e-3))
        model.set_loss_weights(TASK_W, 1)
        out = []
        for d, mk in zip(batches, masks):
            out.append(list(model(d, False, masks=mk)) + [model.grad(n).copy() for n in names])
            opt.step(clip_max_norm=1.0)
        out.append([model.get_parameter(n).copy() for n in names])
        model.close()
        return out
    a, b = run(), run()
    for sa, sb in zip(a, b):
        for x, y in zip(sa, sb):
            assert np.array_equal(np.asarray(x), np.asarray(y))


def test_fp8_row_sharded_table_equals_replicated():
    """the fp8 trunk under the row-sharded item table (world 1): the same losses and trunk gradients as the replicated model"""
    import recommendersystem_amd as ra
    from oracle import synth
    cfg = synth.make_config("f8t", mask_rate=0.2, mask_topk=16)
    rows = 2
    P = synth.make_params(cfg, 13, "test")
    d = synth.make_batch(cfg, rows, 14); mk = synth.make_masks(cfg, rows, 15)
    res = []
    for shard in (None, (0, 1)):
        c = dict(cfg)
        if shard:
            c["table_shard"] = shard
        model = ra.RecommenderModel(c, dtype="fp8", max_rows=rows)
        model.load_state_dict(P)
        model.set_loss_weights(TASK_W, 1)
        losses = model(d, False, masks=mk)
        res.append((losses, {n: model.grad(n).copy() for n in synth.trainable_names(cfg) if "transformers" in n}))
        model.close()
    (la, ga), (lb, gb) = res
    assert np.allclose(la, lb, rtol=2e-3), (la, lb)
    worst = max(float(np.abs(ga[n] - gb[n]).max() / max(np.abs(ga[n]).max(), 1e-6)) for n in ga)
    assert worst < 5e-2, worst


def test_fp8_at_the_benchmark_size_stays_close_to_bf16():
    """cfg-3 at the benchmark's own batch (64 rows x 512; the grouped fp8 weight-gradient launch, grouped-query segments of the fused
    QKV products, 65 536-token casts): the first step's losses and gradient norm from the same weights and batch stay within the fp8
    noise of the bf16 arithmetic, and two steps later the loss has moved the same way"""
    import recommendersystem_amd as ra
    from recommendersystem_amd import workload
    out = {}
    for dtype in ("bf16", "fp8"):
        cfg = workload.make_config("cfg3")
        model = ra.RecommenderModel(cfg, dtype=dtype, max_rows=64)
        model.init_weights(0x1217); model.random_pretrained_embeddings(0x3E7A)
        opt = ra.create_optimizer(model, cfg)
        model.set_loss_weights(ra.make_task_weights(), 1)
        d = workload.make_batch(cfg, 64, 0xD47A, mu=4.6, sigma=1.0)
        losses = [model(d, False)]
        gn = float(np.sqrt(sum(float((model.grad(n).astype(np.float64) ** 2).sum()) for n, _, tr in model.named_parameters() if tr and "transformers.layers" in n)))
        for _ in range(2):
            opt.step(lr_factor=1.0, clip_max_norm=1.0)
            losses.append(model(d, False))
        out[dtype] = (np.array(losses, np.float64), gn)
        model.close()
    (la, ga), (lb, gb) = out["bf16"], out["fp8"]
    print("cfg-3 first-step losses bf16", la[0], "fp8", lb[0], "trunk gradient norm", ga, gb)
    assert np.isfinite(lb).all()
    assert np.abs(lb[0] - la[0]).max() <= 3e-2 * np.abs(la[0]).max(), (la[0], lb[0])
    assert abs(gb - ga) <= 0.15 * ga, (ga, gb)
    assert np.sign(lb[2] - lb[0]).tolist() == np.sign(la[2] - la[0]).tolist() or np.abs(lb[2] - la[2]).max() <= 5e-2 * np.abs(la[2]).max()


def test_fp8_partial_batch_equals_a_model_sized_for_it():
    """a batch with fewer rows than the model was created for (the transposed fp8 copies keep the full batch's row stride, K stops at
    the batch's tokens): same losses and weight gradients as a model created for that many rows"""
    import recommendersystem_amd as ra
    from oracle import synth
    cfg = synth.make_config("f8t", mask_rate=0.2, mask_topk=16)
    P = synth.make_params(cfg, 17, "test")
    d = synth.make_batch(cfg, 3, 18); mk = synth.make_masks(cfg, 3, 19)
    big = synth.make_batch(cfg, 5, 20); mkb = synth.make_masks(cfg, 5, 21)
    names = [n for n in synth.trainable_names(cfg) if "transformers.layers" in n]
    res = []
    for max_rows in (3, 5):
        model = ra.RecommenderModel(cfg, dtype="fp8", max_rows=max_rows)
        model.load_state_dict(P)
        model.set_loss_weights(TASK_W, 1)
        if max_rows == 5:
            model(big, False, masks=mkb)        # a full batch first: stale tokens beyond the partial batch must not leak into K
            model.zero_grad()
        losses = model(d, False, masks=mk)
        res.append((np.array(losses), {n: model.grad(n).copy() for n in names}))
        model.close()
    (la, ga), (lb, gb) = res
    assert np.allclose(la, lb, rtol=1e-5), (la, lb)
    worst = max(float(np.abs(ga[n] - gb[n]).max() / max(np.abs(ga[n]).max(), 1e-9)) for n in names)
    assert worst < 1e-4, worst      # (split-K atomics: summation order only)


def test_fp8_weight_gradients_rounded_to_bf16_before_the_accumulate(monkeypatch):
    """RSYS_F8_DW_ROUND_BF16=1 (what autograd does under autocast -- the product's output tensor is bf16, param.grad fp32):
    each weight-gradient product is summed over all its split-K parts in fp32, rounded to bf16 ONCE and then added to the gradient.
    After one pass from zero gradients every trunk weight gradient is a bf16 number, namely the rounding of the default mode's; a
    second pass without zero_grad adds a second rounded product (gradient accumulation) instead of re-rounding the sum."""
    import recommendersystem_amd as ra
    from oracle import synth
    cfg = synth.make_config("f8t", mask_rate=0.2, mask_topk=16)
    P = synth.make_params(cfg, 23, "test")
    d = synth.make_batch(cfg, 4, 24); mk = synth.make_masks(cfg, 4, 25)
    names = [n for n in synth.trainable_names(cfg) if "transformers.layers" in n]
    mats = [n for n in names if any(k in n for k in ("q_proj", "k_proj", "v_proj", "output_proj", "w1", "w2", "w3"))]
    assert len(mats) == 7 * cfg["num_layers"], mats
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("RSYS_F8_DW_ROUND_BF16", flag)
        model = ra.RecommenderModel(cfg, dtype="fp8", max_rows=4)
        model.load_state_dict(P)
        model.set_loss_weights(TASK_W, 1)
        model(d, False, masks=mk)
        once = {n: model.grad(n).copy() for n in names}
        model(d, False, masks=mk)
        twice = {n: model.grad(n).copy() for n in names}
        res[flag] = (once, twice)
        model.close()
    (g0, g0b), (g1, g1b) = res["0"], res["1"]
    is_bf16 = lambda a: bool(((np.ascontiguousarray(a, np.float32).view(np.uint32) & 0xFFFF) == 0).all())
    for n in mats:
        assert is_bf16(g1[n]), n
        assert not is_bf16(g0[n]), n                       # (so the check above is not vacuous)
        tol = 2.0 ** -8 * np.abs(g0[n]) + 1e-5 * np.abs(g0[n]).max()   # half a bf16 step + the split-K summation order
        assert (np.abs(g1[n] - g0[n]) <= tol).all(), (n, float(np.abs(g1[n] - g0[n]).max()))
        # accumulation: a second rounded product on top of the first (the sum of two bf16 numbers need not be one)
        assert np.abs(g1b[n] - 2 * g1[n]).max() <= 2.0 ** -7 * np.abs(g1[n]).max(), n
    for n in names:
        if n not in mats:                                  # norm weights: not products of the fp8 linears, untouched by the switch
            assert np.allclose(g1[n], g0[n], rtol=1e-4, atol=1e-6 * np.abs(g0[n]).max()), n


def test_fp8_data_parallel_buckets_on_two_concurrent_ranks():
    """the fp8 trunk behind the data-parallel all-reduce with early gradient buckets (in-process rank group): the grouped fp8
    weight-gradient launches run per bucket; every rank ends with the sum of the two ranks' own gradients"""
    import threading

    import recommendersystem_amd as ra
    from oracle import synth
    from recommendersystem_amd import dist as rdist
    cfg = synth.make_config("f8t", mask_rate=0.2, mask_topk=16)
    rows, world = 2, 2
    P = synth.make_params(cfg, 23, "test")
    batches = [synth.make_batch(cfg, rows, 24 + 10 * r) for r in range(world)]
    masks = [synth.make_masks(cfg, rows, 25 + 10 * r) for r in range(world)]
    names = synth.trainable_names(cfg)
    single = []
    for r in range(world):
        model = ra.RecommenderModel(cfg, dtype="fp8", max_rows=rows)
        model.load_state_dict(P); model.set_loss_weights(TASK_W, 1)
        model(batches[r], False, masks=masks[r])
        single.append({n: model.grad(n).astype(np.float64) for n in names})
        model.close()
    want = {n: single[0][n] + single[1][n] for n in names}
    group = rdist.LocalGroup(world)
    out = [None] * world; err = [None] * world

    def rank_fn(r):
        try:
            comm = rdist.LocalComm(group, r)
            model = ra.RecommenderModel(cfg, dtype="fp8", max_rows=rows)
            model.load_state_dict(P); model.set_loss_weights(TASK_W, 1)
            comm.begin_grad_sync(model)
            model(batches[r], False, masks=masks[r])
            comm.all_reduce_grads(model)
            out[r] = {n: model.grad(n).astype(np.float64) for n in names}
            model.close(); comm.close()
        except BaseException as e:   # noqa: BLE001
            err[r] = e
    th = [threading.Thread(target=rank_fn, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    group.close()
    for e in err:
        if e is not None:
            raise e
    for r in range(world):
        worst = max(float(np.abs(out[r][n] - want[n]).max() / max(np.abs(want[n]).max(), 1e-9)) for n in names)
        assert worst < 2e-3, (r, worst)      # (float atomics of the split-K sums: order only)
