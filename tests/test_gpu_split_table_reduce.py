"""GPU: split reduce of the replicated item table's gradient (RecommenderModel.set_split_table_reduce; beyond the reference, opt-in;
DESIGN 7) on concurrent in-process ranks of one GPU against the dense all-reduce: the heads' part of dF summed out of place under the
trunk backward, the batch's token rows gathered as lists in the tail."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TASK_W = [0.05, 0.2, 0.3, 0.25]


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


def _grads(cfg, P, batches, masks, world, split, micro_steps, delay_us=0):
    """every rank: `micro_steps` backward passes over its own batches (the last one with the gradient sync armed), then the reduce"""
    import recommendersystem_amd as ra
    from oracle import synth
    from recommendersystem_amd import dist as rdist
    names = synth.trainable_names(cfg)
    rows = len(masks[0][0][0])
    group = rdist.LocalGroup(world)

    def rank_fn(r):
        comm = rdist.LocalComm(group, r)
        model = ra.RecommenderModel(cfg, dtype="bf16", max_rows=rows)
        model.load_state_dict(P)
        model.set_loss_weights(TASK_W, 1)
        if split:
            model.set_split_table_reduce(True)
        losses = []
        for i in range(micro_steps):
            if i == micro_steps - 1:
                comm.begin_grad_sync(model)
                if delay_us:      # the communicator's stream is busy: the head part's all-reduce starts after the whole backward
                    comm.debug_delay(delay_us)
            losses.append(model(batches[r][i], False, masks=masks[r][i]))
        comm.all_reduce_grads(model)
        early = comm.early_reduced(model)
        sched = comm.grad_schedule(model)
        G = {n: model.grad(n).copy() for n in names}
        model.close(); comm.close()
        return np.array(losses, np.float64), G, early, sched

    res = _run_ranks(world, rank_fn)
    group.close()
    return res


@pytest.mark.parametrize("world,micro_steps", [(2, 1), (4, 2)])
def test_split_table_reduce_equals_the_dense_all_reduce(world, micro_steps):
    """Deterministic mode (two runs are compared): every tensor but the item table goes through the same collectives in both paths and
    must be BITWISE equal; the table's rows are the same sums taken in another order (heads over the ranks first, then the ranks' token
    rows one after the other, against one sum over the ranks of head + tokens): equal to float rounding, on every rank the same bits.
    Ranks draw different batches, so their token rows overlap only partly; with two micro-steps the first one's token rows are part
    of what the early reduce sums."""
    from oracle import synth
    cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16, deterministic=True)
    rows, seed = 2, 57
    P = synth.make_params(cfg, seed, "test")
    batches = [[synth.make_batch(cfg, rows, seed + 1 + 10 * r + i) for i in range(micro_steps)] for r in range(world)]
    masks = [[synth.make_masks(cfg, rows, seed + 2 + 10 * r + i) for i in range(micro_steps)] for r in range(world)]
    names = synth.trainable_names(cfg)
    ref = _grads(cfg, P, batches, masks, world, False, micro_steps)
    got = _grads(cfg, P, batches, masks, world, True, micro_steps)
    table = "item_embedding.matchedid_embedding.embedding.weight"
    assert table in names
    # the tail's gathers carry max_r U_r token rows per rank (rounded up to 64), not the lists' capacity rows * S + 1 (round 6): the
    # schedule's phase-4 entry counts world * rows * (D + 1) gathered floats
    D, S = cfg["embed_dim"], cfg["max_sequence_length"]
    distinct = max(len(np.unique(np.asarray(batches[r][-1]["matchedid"]))) + 1 for r in range(world))
    for r in range(world):
        tail = [hi - lo for lo, hi, ph in got[r][3] if ph == 4]
        assert len(tail) == 1 and tail[0] % (world * (D + 1)) == 0, got[r][3]
        per_rank = tail[0] // (world * (D + 1))
        assert per_rank == min(rows * S + 1, (distinct + 63) // 64 * 64), (per_rank, distinct)
        assert not [1 for lo, hi, ph in ref[r][3] if ph in (3, 4)]
    for r in range(world):
        (la, ga, ea, _), (lb, gb, eb, _) = ref[r], got[r]
        assert np.array_equal(la, lb), (r, la, lb)
        assert eb - ea == ga[table].size, (ea, eb, ga[table].size)      # the split path ran: the table left the tail's dense reduce
        for n in names:
            assert np.array_equal(gb[n], got[0][1][n]), (r, n)          # every rank ends with the same bits
            if n == table:
                scale = float(np.abs(ga[n]).max())
                assert scale > 0
                d = float(np.abs(ga[n] - gb[n]).max())
                assert d <= 2e-6 * scale, (r, n, d, scale)              # another order of the same fp32 additions
            else:
                assert np.array_equal(ga[n], gb[n]), (r, n, float(np.abs(ga[n] - gb[n]).max()))


def test_token_rows_wait_for_a_late_head_reduce():
    """The head part's out-of-place all-reduce READS G[E] on the communicator's stream while the trunk backward later ADDS the batch's
    token rows to G[E] on the model's stream.  With the communicator's stream held back for 0.3 s -- far longer than
    the backward of this tiny model -- a missing cross-stream wait lets the reduce see the token rows, and the tail adds them a second
    time: the table's gradient then differs from the dense path by whole token-row gradients, not by float rounding."""
    from oracle import synth
    cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16, deterministic=True)
    world, rows, seed = 2, 2, 91
    P = synth.make_params(cfg, seed, "test")
    batches = [[synth.make_batch(cfg, rows, seed + 1 + 10 * r)] for r in range(world)]
    masks = [[synth.make_masks(cfg, rows, seed + 2 + 10 * r)] for r in range(world)]
    table = "item_embedding.matchedid_embedding.embedding.weight"
    ref = _grads(cfg, P, batches, masks, world, False, 1)
    got = _grads(cfg, P, batches, masks, world, True, 1, delay_us=300000)
    for r in range(world):
        ga, gb = ref[r][1], got[r][1]
        scale = float(np.abs(ga[table]).max())
        d = float(np.abs(ga[table] - gb[table]).max())
        assert d <= 2e-6 * scale, (r, d, scale)
        for n in ga:
            if n != table:
                assert np.array_equal(ga[n], gb[n]), (r, n)


def test_split_table_reduce_refuses_what_it_does_not_cover():
    import recommendersystem_amd as ra
    from oracle import synth
    m32 = ra.RecommenderModel(synth.make_config("hd64", mask_rate=0.2), dtype="fp32", max_rows=2)
    with pytest.raises(ra.RsysError):      # fp32 parity mode: the metadata-projection gradient reads G[E] itself
        m32.set_split_table_reduce(True)
    m32.close()
    ft = ra.RecommenderModel(synth.make_config("hd64", finetune=True, finetune_metric="rating"), dtype="bf16", max_rows=2)
    with pytest.raises(ra.RsysError):      # finetune: the item table is frozen
        ft.set_split_table_reduce(True)
    ft.close()
