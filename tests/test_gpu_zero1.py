"""GPU: ZeRO-1 (AdamW.enable_zero1; beyond the reference, opt-in) on concurrent in-process ranks of one GPU against the
all-reduce data-parallel path: reduce-scatter of the flat gradient, AdamW on the rank's 1 / world of the parameters with moments for that
part only, all-gather of the parameters."""
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


def _train(cfg, P, batches, masks, world, dtype, zero1, clip, steps):
    import recommendersystem_amd as ra
    from oracle import synth
    from recommendersystem_amd import dist as rdist
    from recommendersystem_amd.optim import AdamW
    names = synth.trainable_names(cfg)
    rows = len(masks[0][0][0])
    group = rdist.LocalGroup(world)

    def rank_fn(r):
        comm = rdist.LocalComm(group, r)
        model = ra.RecommenderModel(cfg, dtype=dtype, max_rows=rows)
        model.load_state_dict(P)
        model.set_loss_weights(TASK_W, 1)
        opt = AdamW(model, lr=1e-2)
        if zero1:
            opt.enable_zero1(comm)
        losses = []
        for i in range(steps):
            losses.append(model(batches[r][i], False, masks=masks[r][i]))
            if not zero1:
                comm.all_reduce_grads(model)
            opt.step(clip_max_norm=clip, grad_div=float(world))
        Pn = {n: model.get_parameter(n).copy() for n in names}
        G = {n: model.grad(n).copy() for n in names}      # (zeroed by the step: every part of it, owned or not)
        opt.close(); model.close(); comm.close()
        return np.array(losses, np.float64), Pn, G

    res = _run_ranks(world, rank_fn)
    group.close()
    return res


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_zero1_equals_the_all_reduce_path(world, dtype):
    """Without gradient clipping (three optimizer steps) the two paths run the same arithmetic on every parameter (the in-process
    group sums in rank order in both collectives): parameters and losses must be BITWISE equal, on every rank.  With the fused global-norm
    clip the norm is a sum of the ranks' partial sums instead of one sum over the buffer: equal to float rounding.  (Deterministic mode:
    two training runs are compared, and the default mode's float atomics alone make two runs differ in the last bit.)"""
    from oracle import synth
    cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16, deterministic=True)   # (bitwise comparisons of two training runs)
    rows, seed = 2, 31
    P = synth.make_params(cfg, seed, "test")
    batches = [[synth.make_batch(cfg, rows, seed + 1 + 10 * r + i) for i in range(3)] for r in range(world)]
    masks = [[synth.make_masks(cfg, rows, seed + 2 + 10 * r + i) for i in range(3)] for r in range(world)]
    names = synth.trainable_names(cfg)
    for clip, steps in ((0.0, 3), (1.0, 1)):
        # clip > 0: ONE step -- the norm is then the sum of the ranks' partial sums, the coefficient may differ in its last bit, and a
        # last-bit difference in the parameters grows from step to step like any perturbation of a training run (measured 1e-7 after
        # one step, 6e-6 after three)
        ref = _train(cfg, P, batches, masks, world, dtype, False, clip, steps)
        z = _train(cfg, P, batches, masks, world, dtype, True, clip, steps)
        for r in range(world):
            (la, pa, _), (lb, pb, gb) = ref[r], z[r]
            assert np.array_equal(la, lb), (clip, r, la, lb)
            for n in names:
                assert not gb[n].any(), (r, n)                                   # no stale gradient anywhere after the step
                assert np.array_equal(pb[n], z[0][1][n]), (r, n)                 # every rank holds the same gathered parameters
                if clip == 0.0:
                    assert np.array_equal(pa[n], pb[n]), (clip, r, n, float(np.abs(pa[n] - pb[n]).max()))
                else:
                    assert np.abs(pa[n] - pb[n]).max() <= 4e-7 * max(np.abs(pa[n]).max(), 1.0), (clip, r, n, float(np.abs(pa[n] - pb[n]).max()))
        assert not np.array_equal(z[0][1][names[0]], P[names[0]])                  # (and the steps did move the parameters)


def test_zero1_refuses_what_it_does_not_cover():
    import recommendersystem_amd as ra
    from oracle import synth
    from recommendersystem_amd import dist as rdist
    from recommendersystem_amd.optim import AdamW
    cfg = synth.make_config("hd64", mask_rate=0.2)
    group = rdist.LocalGroup(1)
    comm = rdist.LocalComm(group, 0)
    model = ra.RecommenderModel(cfg, dtype="bf16", max_rows=2)
    model.init_weights(3)
    opt = AdamW(model, lr=1e-3)
    opt.enable_zero1(comm)
    sd = opt.state_dict()                  # world 1: the rank's part is everything
    assert set(sd["state"]) == {n for n, _, tr in model.named_parameters() if tr}
    opt.close(); model.close()
    ft = ra.RecommenderModel(synth.make_config("hd64", finetune=True, finetune_metric="rating"), dtype="bf16", max_rows=2)
    o2 = AdamW(ft, lr=1e-3)
    with pytest.raises(ra.RsysError):      # finetune (LoRA segment only) keeps the plain optimizer
        o2.enable_zero1(comm)
    o2.close(); ft.close(); comm.close(); group.close()


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_zero1_state_dict_is_the_replicated_optimizers_and_restores(dtype):
    """a ZeRO-1 optimizer could not be checkpointed.  Two steps on two concurrent ranks, then `state_dict(gather=HostGroup)`:
    every rank gets WHOLE moment tensors, bitwise those of the all-reduce run's replicated optimizer (same arithmetic per element,
    deterministic mode).  Loaded into fresh ZeRO-1 optimizers (each keeps its part) and into a fresh replicated one, a third step
    lands on bitwise the same parameters as the uninterrupted runs.  Without `gather` a partitioned optimizer refuses with a clear error."""
    import os, socket
    import recommendersystem_amd as ra
    from oracle import synth
    from recommendersystem_amd import dist as rdist
    from recommendersystem_amd.optim import AdamW
    cfg = synth.make_config("hd64", mask_rate=0.2, mask_topk=16, deterministic=True)
    rows, seed, world = 2, 47, 2
    P = synth.make_params(cfg, seed, "test")
    names = synth.trainable_names(cfg)
    batches = [[synth.make_batch(cfg, rows, seed + 1 + 10 * r + i) for i in range(3)] for r in range(world)]
    masks = [[synth.make_masks(cfg, rows, seed + 2 + 10 * r + i) for i in range(3)] for r in range(world)]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["RSYS_RDZV_PORT"] = str(port)

    def run(zero1, resume_from=None, steps=(0, 1, 2), want_state_after=None):
        group = rdist.LocalGroup(world)

        def rank_fn(r):
            hg = rdist.HostGroup(r, world)
            comm = rdist.LocalComm(group, r)
            model = ra.RecommenderModel(cfg, dtype=dtype, max_rows=rows)
            model.load_state_dict(P if resume_from is None else resume_from[0])
            model.set_loss_weights(TASK_W, 1)
            opt = AdamW(model, lr=1e-2)
            if zero1:
                opt.enable_zero1(comm)
                if resume_from is None and r == 0:
                    with pytest.raises(ValueError, match="partitioned"):
                        opt.state_dict()
            if resume_from is not None:
                opt.load_state_dict(resume_from[1])
            sd = None
            for i in steps:
                model(batches[r][i], False, masks=masks[r][i])
                if not zero1:
                    comm.all_reduce_grads(model)
                opt.step(clip_max_norm=0.0, grad_div=float(world))
                if i == want_state_after:
                    sd = (model.state_dict(), opt.state_dict(gather=hg))
            Pn = {n: model.get_parameter(n).copy() for n in names}
            opt.close(); model.close(); comm.close(); hg.close()
            return Pn, sd

        out = _run_ranks(world, rank_fn)
        group.close()
        return out

    try:
        full_ref = run(False)                                   # three steps, replicated optimizer
        full_z = run(True)
        for n in names:
            assert np.array_equal(full_ref[0][0][n], full_z[0][0][n]), n
        two_ref = run(False, steps=(0, 1), want_state_after=1)
        two_z = run(True, steps=(0, 1), want_state_after=1)
        for r in range(world):
            sa, sb = two_ref[r][1][1], two_z[r][1][1]
            assert sa["step"] == sb["step"] == 2
            for n in names:
                for k in ("exp_avg", "exp_avg_sq"):
                    assert sb["state"][n][k].shape == sa["state"][n][k].shape
                    assert np.array_equal(sa["state"][n][k], sb["state"][n][k]), (r, n, k)
            assert any(np.abs(sb["state"][n]["exp_avg"]).max() > 0 for n in names)
        for zero1 in (True, False):                             # the ZeRO-1 checkpoint resumes either kind of optimizer
            res = run(zero1, resume_from=two_z[0][1], steps=(2,))
            for r in range(world):
                for n in names:
                    assert np.array_equal(res[r][0][n], full_ref[0][0][n]), (zero1, r, n, float(np.abs(res[r][0][n] - full_ref[0][0][n]).max()))
    finally:
        os.environ.pop("RSYS_RDZV_PORT", None)
