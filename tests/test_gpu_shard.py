"""GPU: the row-sharded item table (rows of E / Meta / F and their Adam moments split over ranks,
vocabulary-parallel cross entropy, sparse row exchange).  It is an extension beyond the reference, so there is no
reference oracle for it ("parity unpinned"): the acceptance is equality with the REPLICATED path of this package, which
is pinned to the reference -- at world 1, and at world 2 / 3 with the ranks run as threads of one process on one GPU
(in-process rank group: the same kernels, collectives as device copies; two RCCL ranks cannot share a GPU)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TASK_W = [0.05, 0.2, 0.3, 0.25]
E_NAME = "item_embedding.matchedid_embedding.embedding.weight"
M_NAME = "item_embedding.metadata_embedding.embedding.weight"


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _load(model, P):
    lo, hi = model.table_rows()
    sd = dict(P)
    sd[E_NAME] = P[E_NAME][lo:hi]
    sd[M_NAME] = P[M_NAME][lo:hi]
    model.load_state_dict(sd)
    return lo, hi


def _reference(cfg, P, batches, masks, dtype, lr):
    """replicated table, the ranks' batches as accumulated micro-steps: gradient = sum over ranks, update with grad_div = world"""
    import recommendersystem_amd as ra
    from oracle import synth
    from recommendersystem_amd.optim import AdamW
    rows = len(masks[0][0])
    model = ra.RecommenderModel(cfg, dtype=dtype, max_rows=rows)
    model.load_state_dict(P)
    model.set_loss_weights(TASK_W, 1)
    losses = [model(d, False, masks=mk) for d, mk in zip(batches, masks)]
    names = synth.trainable_names(cfg)
    G = {n: model.grad(n) for n in names}
    opt = AdamW(model, lr=lr)
    opt.step(clip_max_norm=1.0, grad_div=float(len(batches)))
    Pn = {n: model.get_parameter(n) for n in names}
    model.close()
    return losses, G, Pn


def _run_ranks(world, fn):
    out = [None] * world; err = [None] * world

    def body(r):
        try:
            out[r] = fn(r)
        except BaseException as e:   # noqa: BLE001
            err[r] = e
    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize("name,over,rows,seed", [("tiny", dict(mask_rate=0.25, mask_topk=6), 3, 11), ("hd64", dict(mask_rate=0.2, mask_topk=16), 2, 23)])
@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("dtype,tol_loss,tol_grad,tol_par", [("fp32", 2e-5, 2e-4, 2e-5), ("bf16", 2e-2, 5e-2, 2e-3)])
def test_sharded_table_equals_replicated_table(name, over, rows, seed, world, dtype, tol_loss, tol_grad, tol_par):
    import recommendersystem_amd as ra
    from oracle import synth
    from recommendersystem_amd import dist as rdist
    from recommendersystem_amd.optim import AdamW
    cfg = synth.make_config(name, **over)
    P = synth.make_params(cfg, seed, "test")
    batches = [synth.make_batch(cfg, rows, seed + 1 + 10 * r) for r in range(world)]
    masks = [synth.make_masks(cfg, rows, seed + 2 + 10 * r) for r in range(world)]
    lr = 1e-2
    l_ref, G_ref, P_ref = _reference(cfg, P, batches, masks, dtype, lr)
    names = synth.trainable_names(cfg)
    group = rdist.LocalGroup(world) if world > 1 else None

    def rank_fn(r):
        comm = rdist.LocalComm(group, r) if group is not None else None
        c = dict(cfg); c["table_shard"] = (r, world)
        model = ra.RecommenderModel(c, dtype=dtype, max_rows=rows)
        if comm is not None:
            model.set_shard_comm(comm)
        lo, hi = _load(model, P)
        model.set_loss_weights(TASK_W, 1)
        losses = model(batches[r], False, masks=masks[r])
        if comm is not None:
            comm.all_reduce_grads(model)           # dense gradients: summed over ranks; the table rows are left alone
        G = {n: model.grad(n) for n in names}
        opt = AdamW(model, lr=lr)
        opt.step(clip_max_norm=1.0, grad_div=float(world))
        Pn = {n: model.get_parameter(n) for n in names}
        model.close()
        if comm is not None:
            comm.close()
        return losses, G, Pn, (lo, hi)

    res = _run_ranks(world, rank_fn)
    if group is not None:
        group.close()
    V1 = P[E_NAME].shape[0]
    assert [r[3] for r in res] == [(q * V1 // world, (q + 1) * V1 // world) for q in range(world)]
    for r, (losses, G, Pn, (lo, hi)) in enumerate(res):
        for a, b in zip(losses, l_ref[r]):
            assert abs(a - b) <= tol_loss * max(abs(b), 1.0), (r, losses, l_ref[r])
        for n in names:
            ref_g, ref_p = (G_ref[n][lo:hi], P_ref[n][lo:hi]) if n == E_NAME else (G_ref[n], P_ref[n])
            if ref_g.size == 0:
                continue
            scale = max(np.abs(G_ref[n]).max(), 1e-12)
            assert np.abs(G[n] - ref_g).max() <= tol_grad * scale, (r, n, np.abs(G[n] - ref_g).max() / scale)
            # (Adam's first step moves every element by ~lr * sign(g): elements whose gradient is ~0 may differ by 2 lr, so the
            # parameters are compared in the mean, the gradients above element by element)
            assert np.abs(Pn[n] - ref_p).mean() <= tol_par, (r, n, np.abs(Pn[n] - ref_p).mean())


def test_sharded_table_random_init_is_the_replicated_table():
    """init_weights / random_pretrained_embeddings generate the tables by GLOBAL row: the ranks' shards, put together, are
    bit for bit the replicated model's tables (same seed on every rank replaces DDP's rank-0 broadcast)."""
    import recommendersystem_amd as ra
    from oracle import synth
    cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16)
    cfg["vocab_sizes"]["0_matchedid"] = 5000; cfg["vocab_sizes"]["1_matchedid"] = 4321     # more than two 4096-row chunks
    full = ra.RecommenderModel(cfg, dtype="bf16", max_rows=1)
    full.init_weights(7); full.random_pretrained_embeddings(9)
    E = full.get_parameter(E_NAME); M = full.get_parameter(M_NAME)
    other = full.get_parameter("transformers.layers.0.mlp.w1.weight")
    full.close()
    assert not E[-1].any() and not M[-1].any()
    for world in (2, 3):
        for r in range(world):
            c = dict(cfg); c["table_shard"] = (r, world)
            m = ra.RecommenderModel(c, dtype="bf16", max_rows=1)
            m.init_weights(7); m.random_pretrained_embeddings(9)
            lo, hi = m.table_rows()
            assert np.array_equal(m.get_parameter(E_NAME), E[lo:hi]) and np.array_equal(m.get_parameter(M_NAME), M[lo:hi])
            assert np.array_equal(m.get_parameter("transformers.layers.0.mlp.w1.weight"), other)
            m.close()


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-4), ("bf16", 5e-2)])
def test_sharded_table_collectives_through_rccl_world1(monkeypatch, dtype, tol):
    """The production transport: with RCCL forced at world 1 the sharded step's collectives (ncclAllGather of the selected
    rows, ncclAllReduce max / sum of the soft-max statistics and of the selected rows' gradients, grouped ncclSend /
    ncclRecv of the row exchange, the all-reduce of the table rows' squared norm) are really issued -- to the rank itself
    -- and the step still equals the replicated one."""
    import recommendersystem_amd as ra
    from oracle import synth
    from recommendersystem_amd import dist as rdist
    from recommendersystem_amd.optim import clip_grad_norm_
    monkeypatch.setenv("RSYS_FORCE_RCCL", "1")
    cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16)
    rows, seed = 2, 23
    P = synth.make_params(cfg, seed, "test")
    d = synth.make_batch(cfg, rows, seed + 1)
    mk = synth.make_masks(cfg, rows, seed + 2)
    names = synth.trainable_names(cfg)
    ref = ra.RecommenderModel(cfg, dtype=dtype, max_rows=rows)
    ref.load_state_dict(P); ref.set_loss_weights(TASK_W, 1)
    l_ref = ref(d, False, masks=mk)
    G_ref = {n: ref.grad(n) for n in names}
    n_ref = clip_grad_norm_(ref, 1e9)
    ref.close()
    comm = rdist.Comm(rdist.HostGroup(0, 1), 0)
    c = dict(cfg); c["table_shard"] = (0, 1)
    model = ra.RecommenderModel(c, dtype=dtype, max_rows=rows)
    model.set_shard_comm(comm)
    model.load_state_dict(P); model.set_loss_weights(TASK_W, 1)
    losses = model(d, False, masks=mk)
    comm.all_reduce_grads(model)
    for a, b in zip(losses, l_ref):
        assert abs(a - b) <= tol * max(abs(b), 1.0), (losses, l_ref)
    for n in names:
        assert np.abs(model.grad(n) - G_ref[n]).max() <= tol * max(np.abs(G_ref[n]).max(), 1e-12), n
    assert abs(clip_grad_norm_(model, 1e9) - n_ref) <= tol * n_ref
    model.close(); comm.close()


@pytest.mark.parametrize("world", [2, 4])
def test_data_parallel_step_with_early_buckets_on_concurrent_ranks(world):
    """The reference-parity data-parallel scheme (replicated table, DDP bucketed all-reduce, train.py:678-682) on `world`
    CONCURRENT ranks of one GPU (in-process rank group): armed with begin_grad_sync every rank's trunk backward hands its
    finished per-layer gradient buckets to the communicator while it is still running, all_reduce_grads covers the rest
    exactly once, and every rank ends with the SUM of the ranks' gradients and -- after the fused clip + mean + AdamW --
    the parameters of the accumulated single-model step."""
    import recommendersystem_amd as ra
    from oracle import synth
    from recommendersystem_amd import dist as rdist
    from recommendersystem_amd.optim import AdamW
    cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16)
    rows, seed = 2, 23
    P = synth.make_params(cfg, seed, "test")
    batches = [synth.make_batch(cfg, rows, seed + 1 + 10 * r) for r in range(world)]
    masks = [synth.make_masks(cfg, rows, seed + 2 + 10 * r) for r in range(world)]
    l_ref, G_ref, P_ref = _reference(cfg, P, batches, masks, "bf16", 1e-2)
    names = synth.trainable_names(cfg)
    group = rdist.LocalGroup(world)

    def rank_fn(r):
        comm = rdist.LocalComm(group, r)
        comm.self_test()
        assert comm.all_reduce_sum([r + 1.0, 2.0]) == [world * (world + 1) / 2, 2.0 * world]
        model = ra.RecommenderModel(cfg, dtype="bf16", max_rows=rows)
        model.load_state_dict(P)
        model.set_loss_weights(TASK_W, 1)
        comm.begin_grad_sync(model)
        losses = model(batches[r], False, masks=masks[r])
        comm.all_reduce_grads(model)
        early = comm.early_reduced(model)
        G = {n: model.grad(n) for n in names}
        opt = AdamW(model, lr=1e-2)
        opt.step(clip_max_norm=1.0, grad_div=float(world))
        Pn = {n: model.get_parameter(n) for n in names}
        model.close(); comm.close()
        return losses, G, Pn, early

    res = _run_ranks(world, rank_fn)
    group.close()
    D, I, L = cfg["embed_dim"], cfg["intermediate_dim"], cfg["num_layers"]
    nqkv = (cfg["num_heads"] + 2 * cfg["num_kv_heads"]) * (D // cfg["num_heads"])
    for r, (losses, G, Pn, early) in enumerate(res):
        assert early >= L * (nqkv * D + D * D + 2 * I * D + D * I), early       # the trunk's weight matrices went early
        for a, b in zip(losses, l_ref[r]):
            assert abs(a - b) <= 2e-2 * max(abs(b), 1.0)
        for n in names:
            scale = max(np.abs(G_ref[n]).max(), 1e-12)
            assert np.abs(G[n] - G_ref[n]).max() <= 5e-2 * scale, (r, n)
            assert np.array_equal(G[n], res[0][1][n]), (r, n)                     # every rank holds the same reduced bits
            assert np.abs(Pn[n] - P_ref[n]).mean() <= 2e-3, (r, n)


@pytest.mark.parametrize("sharded", [False, True])
def test_replica_consistency_check_on_concurrent_ranks(sharded):
    """SURVEY 2.4 C1 (transformer.py:678-682: DDP broadcasts rank 0's parameters; here: same-seed init + comparison): three concurrent
    ranks of one GPU, device-side checksum words (`rsys_param_checksum`) gathered through the communicator's fp64 all-reduce.  Equal after
    the same-seed init and after one data-parallel optimizer step on different batches (the all-reduce keeps replicas equal); the lowest
    mantissa bit of ONE norm scale changed on rank 2 fails the check on EVERY rank.  Row-sharded table: the ranks' own table rows differ
    by construction and stay out of the words; everything replicated is still compared."""
    import recommendersystem_amd as ra
    from oracle import synth
    from recommendersystem_amd import dist as rdist
    from recommendersystem_amd.optim import AdamW
    world, rows = 3, 2
    cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16)
    batches = [synth.make_batch(cfg, rows, 71 + 10 * r) for r in range(world)]
    masks = [synth.make_masks(cfg, rows, 72 + 10 * r) for r in range(world)]
    group = rdist.LocalGroup(world)

    def rank_fn(r):
        comm = rdist.LocalComm(group, r)
        c = dict(cfg)
        if sharded:
            c["table_shard"] = (r, world)
        model = ra.RecommenderModel(c, dtype="bf16", max_rows=rows)
        model.init_weights(0x1217); model.random_pretrained_embeddings(0x3E7A)
        if sharded:
            model.set_shard_comm(comm)
        w0 = rdist.assert_replicas_equal(model, comm, "after init")
        assert w0 == model.param_checksum() and np.isfinite(w0).all() and w0[1] > 0
        model.set_loss_weights(TASK_W, 1)
        if not sharded:
            comm.begin_grad_sync(model)
        model(batches[r], False, masks=masks[r])
        comm.all_reduce_grads(model)
        AdamW(model, lr=1e-2).step(clip_max_norm=1.0, grad_div=float(world))
        w1 = rdist.assert_replicas_equal(model, comm, "after one step")
        assert w1 != w0
        name = "transformers.layers.1.mlp_norm.scale"
        if r == 2:
            v = model.get_parameter(name)
            v.view(np.uint32)[5] ^= 1
            model.set_parameter(name, v)
        try:
            rdist.assert_replicas_equal(model, comm, "after the bit flip")
            err = None
        except rdist.ReplicaMismatch as e:
            err = str(e)
        model.close(); comm.close()
        return w0, w1, err

    res = _run_ranks(world, rank_fn)
    group.close()
    assert res[0][0] == res[1][0] == res[2][0] and res[0][1] == res[1][1] == res[2][1]
    for w0, w1, err in res:
        assert err is not None and "ranks [2] do not hold rank 0's parameters" in err, err


@pytest.mark.parametrize("world", [1, 2])
def test_sampled_softmax_with_every_class_sampled_is_the_full_softmax(world):
    """cfg-4's sampled soft-max has no reference counterpart; its machinery (sampled-class gather, target logit by the
    target's owner, importance-weighted partition sum, sparse gradient rows) is pinned by the degenerate case: with at least
    as many samples as local classes every class is drawn once with q = 1, and the step must equal the replicated
    full-soft-max step."""
    import recommendersystem_amd as ra
    from oracle import synth
    from recommendersystem_amd import dist as rdist
    cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16)
    rows, seed = 2, 23
    P = synth.make_params(cfg, seed, "test")
    batches = [synth.make_batch(cfg, rows, seed + 1 + 10 * r) for r in range(world)]
    masks = [synth.make_masks(cfg, rows, seed + 2 + 10 * r) for r in range(world)]
    l_ref, G_ref, _ = _reference(cfg, P, batches, masks, "fp32", 1e-2)
    names = synth.trainable_names(cfg)
    group = rdist.LocalGroup(world) if world > 1 else None

    def rank_fn(r):
        comm = rdist.LocalComm(group, r) if group is not None else None
        c = dict(cfg); c["table_shard"] = (r, world); c["sampled_softmax"] = 1 << 20
        model = ra.RecommenderModel(c, dtype="fp32", max_rows=rows)
        if comm is not None:
            model.set_shard_comm(comm)
        lo, hi = _load(model, P)
        model.set_loss_weights(TASK_W, 1)
        losses = model(batches[r], False, masks=masks[r])
        if comm is not None:
            comm.all_reduce_grads(model)
        G = {n: model.grad(n) for n in names}
        model.close()
        if comm is not None:
            comm.close()
        return losses, G, (lo, hi)

    res = _run_ranks(world, rank_fn)
    if group is not None:
        group.close()
    for r, (losses, G, (lo, hi)) in enumerate(res):
        for a, b in zip(losses, l_ref[r]):
            assert abs(a - b) <= 2e-5 * max(abs(b), 1.0), (r, losses, l_ref[r])
        for n in names:
            ref_g = G_ref[n][lo:hi] if n == E_NAME else G_ref[n]
            assert np.abs(G[n] - ref_g).max() <= 2e-4 * max(np.abs(G_ref[n]).max(), 1e-12), (r, n)


def test_sampled_softmax_estimates_the_full_loss():
    """With fewer samples than classes the loss is an importance-weighted estimate of the full soft-max loss: fresh samples
    every step, the mean over steps within a few percent of the full loss, gradients finite, and only the sampled / target /
    token rows of the table gradient non-zero (what makes a sparse reduce of it possible)."""
    import recommendersystem_amd as ra
    from oracle import synth
    cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16)
    cfg["vocab_sizes"]["0_matchedid"] = 3000; cfg["vocab_sizes"]["1_matchedid"] = 2000
    rows = 4
    d = synth.make_batch(cfg, rows, 31)
    mk = synth.make_masks(cfg, rows, 32)
    full = ra.RecommenderModel(dict(cfg, table_shard=(0, 1)), dtype="fp32", max_rows=rows)
    full.init_weights(5); full.random_pretrained_embeddings(6)
    full.set_loss_weights(TASK_W, 1)
    l_full = full(d, False, masks=mk)
    P = full.state_dict()
    full.close()
    samp = ra.RecommenderModel(dict(cfg, table_shard=(0, 1), sampled_softmax=256), dtype="fp32", max_rows=rows)
    samp.load_state_dict({k: v for k, v in P.items() if not k.startswith("watch_head.")})
    samp.set_loss_weights(TASK_W, 1)
    samp.upload(d, mk)
    seen = []
    for step in range(24):
        samp.zero_grad()
        samp.forward_resident(False, step=step)
        seen.append(samp.losses(False))
    seen = np.array(seen)
    assert len({tuple(np.round(x, 6)) for x in seen[:, [0, 2]]}) > 20           # fresh samples every step
    for ti in (0, 2):
        assert abs(seen[:, ti].mean() - l_full[ti]) <= 0.05 * l_full[ti], (ti, seen[:, ti].mean(), l_full[ti])
    assert np.allclose(seen[:, 1], l_full[1], rtol=1e-5) and np.allclose(seen[:, 3], l_full[3], rtol=1e-5)   # rating heads untouched
    gE = samp.grad(E_NAME)
    assert np.isfinite(gE).all()
    touched = int((np.abs(gE).sum(axis=1) > 0).sum())
    assert 0 < touched <= 2 * 256 + 2 * cfg["mask_topk"] * rows + rows * cfg["max_sequence_length"] + 1
    samp.close()


def test_sampled_softmax_is_unbiased_on_a_peaked_model_and_evaluates_exactly():
    """The estimator where it matters: logits far from flat (O(1) embeddings), 8 % of the classes sampled.  The mean of many
    sampled-soft-max gradients (fresh strata draws, in-batch targets always present with weight 1, sampled classes weighted by
    their stratum's size) must point where the full soft-max gradient points; a global 1/q weight over strata of unequal size,
    or target classes that are only pushed down when they happen to be sampled, fail this (and make a training run drift:
    tools/converge_sampled.py).  An evaluation pass of a sampled-soft-max model reports the exact full soft-max loss."""
    import recommendersystem_amd as ra
    from oracle import synth
    cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16)
    cfg["vocab_sizes"]["0_matchedid"] = 3000; cfg["vocab_sizes"]["1_matchedid"] = 2000
    rows, seed = 4, 41
    P = synth.make_params(cfg, seed, "test")
    for k in (E_NAME, "item_embedding.projection_layer.weight", "item_embedding.projection_layer.bias"):
        P[k] = P[k] * np.float32(0.3)      # logits with a spread of ~2.5: full loss 11.1 / 9.7 against ln 3000 = 8.0
    d = synth.make_batch(cfg, rows, seed + 1)
    mk = synth.make_masks(cfg, rows, seed + 2)
    names = [E_NAME, "transformers.layers.0.mlp.w1.weight", "item_embedding.projection_layer.weight"]
    full = ra.RecommenderModel(dict(cfg, table_shard=(0, 1)), dtype="fp32", max_rows=rows)
    _load(full, P)
    full.set_loss_weights(TASK_W, 1)
    l_full = full(d, False, masks=mk)
    g_full = {n: full.grad(n).astype(np.float64) for n in names}
    e_full = full(d, True, masks=mk)
    full.close()
    samp = ra.RecommenderModel(dict(cfg, table_shard=(0, 1), sampled_softmax=240), dtype="fp32", max_rows=rows)
    _load(samp, P)
    samp.set_loss_weights(TASK_W, 1)
    e_samp = samp(d, True, masks=mk)
    assert np.allclose([e_samp[0], e_samp[2]], [e_full[0], e_full[2]], rtol=1e-5), (e_samp, e_full)     # evaluation: exact loss
    samp.upload(d, mk)
    reps, acc, ls = 96, {n: 0.0 for n in names}, []
    for r in range(reps):
        samp.zero_grad()
        samp.forward_resident(False, step=500 + r)
        ls.append(samp.losses(False))
        for n in names:
            acc[n] = acc[n] + samp.grad(n).astype(np.float64)
    ls = np.array(ls)
    for ti in (0, 2):     # E[log Z^] <= log Z: the estimate sits slightly below the full loss, by much less than its spread
        assert abs(ls[:, ti].mean() - l_full[ti]) <= 0.04 * abs(l_full[ti]) + 3 * ls[:, ti].std() / np.sqrt(reps), (ti, ls[:, ti].mean(), l_full[ti])
    for n in names:
        mean = acc[n] / reps
        cos = float((mean * g_full[n]).sum() / (np.linalg.norm(mean) * np.linalg.norm(g_full[n]) + 1e-30))
        ratio = float(np.linalg.norm(mean) / np.linalg.norm(g_full[n]))
        assert cos > 0.99 and 0.93 < ratio < 1.07, (n, cos, ratio)     # measured: cos 0.998, ratio 0.985-0.995
    samp.close()


def test_sharded_checkpoint_is_the_reference_layout_and_reshards(tmp_path):
    """Checkpoints of a row-sharded run hold the WHOLE tables (the reference's state-dict layout): the ranks' rows and Adam
    moments are gathered over the control plane, one rank writes; the file loads into a replicated model and into a run with
    a different number of shards (each rank keeps its rows)."""
    import socket

    import recommendersystem_amd as ra
    from oracle import synth
    from recommendersystem_amd import dist as rdist
    from recommendersystem_amd import train as T
    from recommendersystem_amd.optim import AdamW
    cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16)
    rows, seed, world = 2, 23, 2
    P = synth.make_params(cfg, seed, "test")
    batches = [synth.make_batch(cfg, rows, seed + 1 + 10 * r) for r in range(world)]
    masks = [synth.make_masks(cfg, rows, seed + 2 + 10 * r) for r in range(world)]
    _, _, P_ref = _reference(cfg, P, batches, masks, "fp32", 1e-2)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    import os
    os.environ["RSYS_RDZV_PORT"] = str(port)
    group = rdist.LocalGroup(world)
    sched = T.LambdaLR(T.WSDScheduler(warmup_steps=2, total_steps=40, decay_ratio=0.1, final_ratio=0.1))

    def rank_fn(r):
        hg = rdist.HostGroup(r, world)
        comm = rdist.LocalComm(group, r)
        c = dict(cfg); c["table_shard"] = (r, world)
        model = ra.RecommenderModel(c, dtype="fp32", max_rows=rows)
        model.set_shard_comm(comm)
        _load(model, P)
        model.set_loss_weights(TASK_W, 1)
        model(batches[r], False, masks=masks[r])
        comm.all_reduce_grads(model)
        opt = AdamW(model, lr=1e-2)
        opt.step(clip_max_norm=1.0, grad_div=float(world))
        T.checkpoint_model(str(tmp_path), model, opt, sched, cfg, 0, [1.0] * 4, [1.0] * 4, TASK_W, True, gather=hg, write=r == 0)
        model.close(); comm.close(); hg.close()

    _run_ranks(world, rank_fn)
    group.close()
    os.environ.pop("RSYS_RDZV_PORT", None)
    z = np.load(tmp_path / "transformer.masked.npz")
    V1 = P[E_NAME].shape[0]
    assert z["model/" + E_NAME].shape == (V1, cfg["embed_dim"]) and z["optimizer/exp_avg/" + E_NAME].shape == (V1, cfg["embed_dim"])
    assert np.abs(z["model/" + E_NAME] - P_ref[E_NAME]).mean() <= 2e-5               # = the replicated run's table after the step
    assert np.abs(z["model/transformers.layers.0.mlp.w1.weight"] - P_ref["transformers.layers.0.mlp.w1.weight"]).mean() <= 2e-5
    # resume into three shards and into a replicated model: every holder gets exactly its rows
    for shard in (None, (0, 3), (2, 3)):
        c = dict(cfg)
        if shard:
            c["table_shard"] = shard
        m = ra.RecommenderModel(c, dtype="fp32", max_rows=rows)
        P0 = {k: v for k, v in P.items()}
        m.load_state_dict(P0)                                       # (frozen table etc.)
        opt = AdamW(m, lr=1e-2)
        epoch, _ = T.load_checkpoint(str(tmp_path / "transformer.masked.npz"), m, opt, None)
        lo, hi = m.table_rows()
        assert epoch == 0 and np.array_equal(m.get_parameter(E_NAME), z["model/" + E_NAME][lo:hi])
        assert np.array_equal(opt.state_dict()["state"][E_NAME]["exp_avg_sq"], z["optimizer/exp_avg_sq/" + E_NAME][lo:hi])
        m.close()


def test_sharded_step_enqueues_without_draining_the_stream():
    """the vocabulary-parallel heads used to drain the stream once per watch task to learn the sizes of their
    collectives.  Those sizes depend on the masked batch only: they are gathered and copied to the host BEFORE the trunk forward,
    and the heads wait for that (by then long signalled) event once.  Counted inside the C ABI: from the second step on a
    training pass of a sharded model takes 0 stream drains and 1 event wait, also with the sampled soft-max."""
    import recommendersystem_amd as ra
    from oracle import synth
    from recommendersystem_amd import dist as rdist
    cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16)
    world, rows = 2, 3
    P = synth.make_params(cfg, 23, "test")
    batches = [synth.make_batch(cfg, rows, 24 + 10 * r) for r in range(world)]
    for sampled in (0, 16):
        group = rdist.LocalGroup(world)

        def rank_fn(r):
            comm = rdist.LocalComm(group, r)
            c = dict(cfg); c["table_shard"] = (r, world)
            if sampled:
                c["sampled_softmax"] = sampled
            model = ra.RecommenderModel(c, dtype="bf16", max_rows=rows)
            model.set_shard_comm(comm)
            _load(model, P)
            model.set_loss_weights(TASK_W, 1)
            counts = []
            for step in range(3):
                losses = model(batches[r], False)
                counts.append(model.debug_get("host_syncs", rows).tolist())
                assert np.isfinite(losses).all()
            model.close(); comm.close()
            return counts
        res = _run_ranks(world, rank_fn)
        group.close()
        for counts in res:
            assert counts[1] == [0, 1] and counts[2] == [0, 1], (sampled, counts)
