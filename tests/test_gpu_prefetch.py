"""GPU: the pipelined host loop (rsys_batch_prefetch / rsys_batch_swap: the next batch checked, packed and copied beside the running
step -- the reference's DataLoader workers + non_blocking to_device, train.py:162-165,178-184) against the plain loop that uploads
each batch after the previous step: same losses, same parameters, bit for bit; a rejected prefetch leaves the resident batch alone."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _epoch(pipelined, accum, n_batches=7):
    import recommendersystem_amd as ra
    from oracle import synth
    from recommendersystem_amd.train import ConstantScheduler, LambdaLR, make_task_weights, train_epoch
    cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16, deterministic=True)
    rows = 3
    P = synth.make_params(cfg, 41, "test")
    batches = [synth.make_batch(cfg, rows if i % 3 else rows - 1, 100 + i) for i in range(n_batches)]   # (row counts differ between batches)
    model = ra.RecommenderModel(cfg, dtype="bf16", max_rows=rows)
    model.load_state_dict(P)
    model.mask_seed = 77
    opt = ra.create_optimizer(model, cfg)
    model._no_prefetch = not pipelined
    assert model.can_prefetch == pipelined
    loss = train_epoch(model, batches, opt, LambdaLR(ConstantScheduler()), make_task_weights(), accum, None)
    names = synth.trainable_names(cfg)
    out = {n: model.get_parameter(n).copy() for n in names}
    model.close()
    return np.asarray(loss, np.float64), out


@pytest.mark.parametrize("accum", [1, 2])
def test_pipelined_loop_equals_the_plain_loop_bit_for_bit(accum):
    l0, p0 = _epoch(False, accum)
    l1, p1 = _epoch(True, accum)
    assert np.array_equal(l0, l1), (l0, l1)
    for n in p0:
        assert np.array_equal(p0[n], p1[n]), n


def test_rejected_prefetch_leaves_the_resident_batch_and_swap_needs_a_prefetch():
    import recommendersystem_amd as ra
    from oracle import synth
    cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16, deterministic=True)
    model = ra.RecommenderModel(cfg, dtype="bf16", max_rows=2)
    model.load_state_dict(synth.make_params(cfg, 5, "test"))
    model.set_loss_weights([0.05, 0.2, 0.3, 0.25], 1)
    model.mask_seed = 9
    good = synth.make_batch(cfg, 2, 11)
    with pytest.raises(ra.RsysError):
        model.swap_batch()                              # nothing prefetched yet
    model.upload(good)
    model.forward_resident(False, step=0)
    ref = model.losses(False)
    bad = {k: np.array(v).copy() for k, v in synth.make_batch(cfg, 2, 12).items()}
    bad["matchedid"][3] = 10 ** 7                        # out of range: caught on the host before anything is copied
    with pytest.raises(ra.RsysError):
        model.prefetch(bad)
    with pytest.raises(ra.RsysError):
        model.swap_batch()                              # the rejected batch is not pending
    model.zero_grad()
    model.forward_resident(False, step=0)
    assert model.losses(False) == ref                   # the resident batch is untouched
    # a prefetch that is superseded by an explicit upload is dropped
    model.prefetch(synth.make_batch(cfg, 2, 13))
    model.upload(good)
    with pytest.raises(ra.RsysError):
        model.swap_batch()
    model.close()


def test_parked_losses_are_the_step_by_step_ones():
    """rsys_losses_push / _drain (the reference adds its losses into device tensors and reads them at the end of the epoch,
    transformer.py:245-262): three steps parked and read once give exactly what rsys_losses_get gave after each step; an empty drain is
    an empty list."""
    import recommendersystem_amd as ra
    from oracle import synth
    cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16, deterministic=True)
    batches = [synth.make_batch(cfg, 2, 300 + i) for i in range(3)]

    def run(parked):
        model = ra.RecommenderModel(cfg, dtype="bf16", max_rows=2)
        model.load_state_dict(synth.make_params(cfg, 6, "test"))
        model.set_loss_weights([0.05, 0.2, 0.3, 0.25], 1)
        model.mask_seed = 13
        opt = ra.create_optimizer(model, cfg)
        out = []
        assert model.drain_losses() == []
        for i, b in enumerate(batches):
            model.upload(b)
            model.forward_resident(False, step=i)
            opt.step(clip_max_norm=1.0)
            if parked:
                model.push_losses()
            else:
                lo = model.losses(False)
                out.append((lo, list(model.last_weight_sums)))
        if parked:
            out = model.drain_losses()
            assert model.drain_losses() == []
        model.close()
        return out
    a, b = run(False), run(True)
    assert len(a) == len(b) == 3
    for (la, wa), (lb, wb) in zip(a, b):
        assert la == lb and wa == wb, (la, lb, wa, wb)
    assert a[0][0] != a[1][0]


def test_an_epoch_that_left_by_an_exception_leaves_no_parked_step_to_the_next(tmp_path):
    """train_epoch parks every step's losses on the device and reads them every LOSS_RING steps; an epoch that leaves by an
    exception between a push and the drain (here: a batch the prefetch rejects, four steps in) used to hand its parked steps to the NEXT
    epoch's mean.  Now an epoch starts by emptying the ring: the epoch after the failed one reports exactly what it reports on a fresh
    model brought to the same state."""
    import recommendersystem_amd as ra
    from oracle import synth
    from recommendersystem_amd.train import ConstantScheduler, LambdaLR, make_task_weights, train_epoch
    cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16, deterministic=True)
    rows = 2
    good = [synth.make_batch(cfg, rows, 500 + i) for i in range(4)]
    bad = {k: np.array(v).copy() for k, v in synth.make_batch(cfg, rows, 600).items()}
    bad["matchedid"][1] = 10 ** 7
    second = [synth.make_batch(cfg, rows, 700 + i) for i in range(3)]
    tw = make_task_weights()

    def run(fail):
        model = ra.RecommenderModel(cfg, dtype="bf16", max_rows=rows)
        model.load_state_dict(synth.make_params(cfg, 8, "test"))
        model.mask_seed = 21
        opt = ra.create_optimizer(model, cfg)
        sched = LambdaLR(ConstantScheduler())
        if fail:
            with pytest.raises(ra.RsysError):
                train_epoch(model, good + [bad], opt, sched, tw, 1, None)
        else:
            train_epoch(model, good, opt, sched, tw, 1, None)
        out = train_epoch(model, second, opt, sched, tw, 1, None)
        model.close()
        return np.asarray(out, np.float64)
    a, b = run(False), run(True)
    assert np.array_equal(a, b), (a, b)
