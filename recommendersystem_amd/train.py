"""Host-side training-loop logic, mirroring notebooks/Training/transformer.py ("train.py").

Same names, argument meaning and behaviour as the reference functions; each cites
the lines it restates.  Arithmetic on the device goes through RecommenderModel /
AdamW (C ABI); nothing here touches a CPU compute fallback.
"""
import dataclasses
import math

import numpy as np

from .model import ALL_MEDIUMS, ALL_METRICS
from .optim import clip_grad_norm_


class ConstantScheduler:
    """Finetune schedule (train.py:301-307): the factor is 1 on every call; `steps` counts the calls."""
    steps = 0

    def __call__(self, _step=None):
        self.steps += 1
        return 1


class WSDScheduler:
    """Warm-up / stable / decay learning-rate factor (train.py:310-328) as a piecewise-linear curve through the knots
    (0, 0) - (warmup, 1) - (total - decay, 1) - (total, final_ratio), decay = int(total * decay_ratio); steps outside
    [0, total] take the value of the nearest end."""

    def __init__(self, warmup_steps, total_steps, decay_ratio, final_ratio):
        decay = int(total_steps * decay_ratio)
        plateau_end = total_steps - decay
        if plateau_end < warmup_steps:
            raise AssertionError("warm-up and decay overlap: total_steps is too small")
        self.warmup_steps, self.total_steps, self.final_ratio = warmup_steps, total_steps, final_ratio
        # (first step, length, factor at the start, factor at the end) of the two ramps
        self._ramps = ((0, max(1, warmup_steps), 0.0, 1.0), (plateau_end, max(1, decay), 1.0, final_ratio))

    def __call__(self, step):
        s = min(max(int(step), 0), self.total_steps)
        (w0, wn, wa, wb), (d0, dn, da, db) = self._ramps
        if s <= self.warmup_steps:
            return wa + (wb - wa) * ((s - w0) / wn) if self.warmup_steps else 0.0
        if s <= d0:
            return 1.0
        return da + (db - da) * ((s - d0) / dn)

    def reference_state(self):
        """The attribute dict of the reference's scheduler object (what torch's LambdaLR.state_dict() stores in
        `lr_lambdas`; checkpoint interchange, checkpoint.py)."""
        d0 = self._ramps[1][0]
        return {"warmup_steps": self.warmup_steps, "total_steps": self.total_steps, "final_ratio": self.final_ratio,
                "decay_steps": self.total_steps - d0, "stable_steps": d0 - self.warmup_steps}


class LambdaLR:
    """torch.optim.lr_scheduler.LambdaLR as train.py:333,347 uses it: factor(step), stepped once per optimizer step."""

    def __init__(self, fn):
        self.fn = fn
        self.last_epoch = 0
        self._factor = fn(0)

    def step(self):
        self.last_epoch += 1
        self._factor = self.fn(self.last_epoch)

    def factor(self):
        return self._factor

    def state_dict(self):
        return {"last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.last_epoch = sd["last_epoch"]
        self._factor = self.fn(self.last_epoch)


def create_learning_rate_schedule(tokens_per_epoch, tokens_per_batch, epochs, finetune=False, warmup_steps=2000):
    """train.py:331-347 (the reference's warm-up is fixed at 2000 steps, which a run needs at least 2223 steps for)."""
    if finetune:
        return LambdaLR(ConstantScheduler())
    total_steps = int(round(tokens_per_epoch * epochs / tokens_per_batch))
    return LambdaLR(WSDScheduler(warmup_steps=warmup_steps, total_steps=total_steps, decay_ratio=0.1, final_ratio=0.1))


@dataclasses.dataclass
class EarlyStopper:
    """Two running minima over the epoch scores (train.py:350-372).  `early_stop` turns on once `patience` epochs in a
    row failed to beat the best score by the relative margin `rtol`; `save_model` says whether the latest score is the
    lowest seen so far (the checkpoint is written only then)."""
    patience: float
    rtol: float
    counter: int = 0                      # consecutive epochs without a margin-beating score
    early_stop: bool = False
    save_model: bool = False
    _best_margin: float = math.inf        # best score for the patience rule
    _best: float = math.inf               # best score for checkpointing

    def __call__(self, score):
        improved = score < self._best_margin * (1 - self.rtol)
        self.counter = 0 if improved else self.counter + 1
        if improved:
            self._best_margin = score
        elif self.counter >= self.patience:
            self.early_stop = True
        self.save_model = score < self._best
        self._best = min(self._best, score)


def make_early_stopper(config):
    """train.py:409-413: finetune runs stop after two flat epochs (0.1 % margin), pretraining never stops early."""
    return EarlyStopper(2, 0.001) if config["finetune"] else EarlyStopper(math.inf, 0)


def wsum(values, weights):
    """Weighted total of the per-task losses (train.py:375-376)."""
    return sum(v * w for v, w in zip(values, weights))


# per-task loss scales of train.py:381-390, order ALL_MEDIUMS x ALL_METRICS
_TASK_SCALE = np.array([4.618602403897067, 1.1958987168236102, 2.5443243303769867, 1.0527565486045412])


def make_task_weights(finetune_medium=None, finetune_metric=None):
    """train.py:379-406 (args.finetune_medium / args.finetune_metric become parameters): the outer product of a medium
    share and a metric share, normalised to sum 1, each entry divided by its task's loss scale.  Pretraining: manga
    0.25 / anime 1, watch 1 / rating 0.25; finetuning: one-hot on the chosen medium and metric."""
    if finetune_medium is None:
        medium_share = np.array([0.25, 1.0])
    else:
        medium_share = np.eye(len(ALL_MEDIUMS))[ALL_MEDIUMS.index(finetune_medium)]
    if finetune_metric is None:
        metric_share = np.array([1.0, 0.25])
    else:
        metric_share = np.eye(len(ALL_METRICS))[ALL_METRICS.index(finetune_metric)]
    share = np.outer(medium_share, metric_share).ravel()
    return [float(x) for x in share / share.sum() / _TASK_SCALE]


def minimize_quadratic(x, y):
    """Value at the vertex of the parabola through three points (train.py:187-196), by Newton's divided differences:
    p(t) = y0 + d01 (t - x0) + c (t - x0)(t - x1); flat data returns that constant."""
    (x0, x1, x2), (y0, y1, y2) = x, y
    if y0 == y1 == y2:
        return float(y0)
    d01 = (y1 - y0) / (x1 - x0)
    d12 = (y2 - y1) / (x2 - x1)
    c = (d12 - d01) / (x2 - x0)
    t = (x0 + x1) / 2 - d01 / (2 * c)
    return float(y0 + d01 * (t - x0) + c * (t - x0) * (t - x1))


def reduce_mean(comm, x, w):
    """train.py:199-204: two SUM all-reduces, then the weighted mean per task."""
    x = [float(v) for v in x]; w = [float(v) for v in w]
    if comm is not None:
        tot = comm.all_reduce_sum(x + w)
        x, w = tot[: len(x)], tot[len(x):]
    return [a / b if b != 0 else 0 for (a, b) in zip(x, w)]


_RATING_SCALES = [1, 0, -1]   # prediction scales of the three evaluation moments (model.py:395-401)


def lockstep_batches(model, dataloader, comm):
    """The batches of `dataloader`, in step with the other ranks when the item table is row-sharded (config["table_shard"]).
    There every upload / forward contains collectives (exchange plan, row exchange, vocabulary-parallel heads), so a rank that
    had one batch more or fewer than its peers would leave them waiting inside a collective; the reference's forward has none,
    and its per-rank loaders may well differ in length (train.py:46-51: shards by file).  Before every batch the ranks add up
    who still has one (one scalar all-reduce over the communicator) and ALL stop at the first iteration in which somebody has
    run out; the longer loaders' tails are dropped.  Replicated table: the loader as it is."""
    sharded = comm is not None and comm.world > 1 and bool(getattr(model, "config", {}).get("table_shard"))
    if not sharded:
        yield from dataloader
        return
    it = iter(dataloader)
    while True:
        batch = next(it, None)
        ready = comm.all_reduce_sum([0.0 if batch is None else 1.0])[0]
        if ready < comm.world - 0.5:
            return
        yield batch


def evaluate_metrics(model, dataloader, comm=None):
    """train.py:207-235: eval forward (fresh random masks, model(d, True)), weight-averaged per task,
    rating -> quadratic-minimum MSE over prediction scales {1, 0, -1}."""
    tasks = [(m, metric) for m in ALL_MEDIUMS for metric in ALL_METRICS]
    moments = np.zeros((len(tasks), len(_RATING_SCALES)))   # watch tasks use column 0 only
    mass = np.zeros(len(tasks))
    model.eval()
    for batch in lockstep_batches(model, dataloader, comm):
        out = model(batch, True)
        w = np.asarray(model.last_weight_sums, np.float64)
        for i, (_, metric) in enumerate(tasks):
            vals = out[i] if metric == "rating" else [out[i]]
            if w[i] != 0:
                moments[i, : len(vals)] += w[i] * np.asarray(vals, np.float64)
        mass += w
    model.train()
    totals = [minimize_quadratic(_RATING_SCALES, list(moments[i])) if metric == "rating" else float(moments[i, 0])
              for i, (_, metric) in enumerate(tasks)]
    return reduce_mean(comm, totals, mass)


def train_epoch(model, dataloader, optimizer, scheduler, task_weights, grad_accum_steps, comm=None, max_norm=1.0):
    """train.py:238-283.  Micro-steps accumulate locally (DDP no_sync, :268-271); the last one is followed by the
    gradient all-reduce (DDP hook, :272), clip_grad_norm_(1.0) (:273), optimizer.step + zero_grad (:274-275) and
    scheduler.step (:276).  Clip + mean + AdamW run as one fused device pass."""
    n_tasks = len(task_weights)
    training_losses = [0.0] * n_tasks
    training_weights = [0.0] * n_tasks
    model.set_loss_weights(task_weights, grad_accum_steps)
    optimizer.zero_grad(set_to_none=True)
    world = 1 if comm is None else comm.world
    # The loop's one host sync per step is the loss read-back (float(loss), train.py:261-263).  Everything the host does for the NEXT
    # batch -- taking it from the loader, the index checks, packing and the copy to the device -- is placed between enqueueing this
    # step's optimizer pass and that read-back (model.prefetch / swap_batch), so it overlaps the step on the device instead of
    # following it; the arithmetic and the order of the sums are those of the plain loop.  (Row-sharded table: plain uploads.)
    staged = all(hasattr(model, a) for a in ("upload", "forward_resident", "losses"))   # (a model that is only callable: the plain loop)
    pipelined = staged and bool(getattr(model, "can_prefetch", False))
    # ... and the read-back itself is deferred the way the reference defers it (its losses are device tensors summed on the device and
    # read at the end of the epoch, transformer.py:245-262, 279-283): each step's sums are parked in a device ring and read every
    # LOSS_RING steps, in order, with the same host arithmetic -- the same floats as the step-by-step read, without a host wait per step.
    deferred = pipelined and hasattr(model, "push_losses")
    parked = 0
    if deferred:
        model.drain_losses()                   # steps an earlier epoch parked before it left by an exception are not this epoch's
    it = iter(lockstep_batches(model, dataloader, comm))
    data = next(it, None)
    if data is not None and staged:
        model.upload(data)
    step = 0
    while data is not None:
        zero1 = getattr(optimizer, "_zero1", None) is not None   # (opt-in ZeRO-1: the optimizer step reduces the gradient itself)
        last_micro = (step + 1) % grad_accum_steps == 0
        if comm is not None and not zero1 and last_micro:
            comm.begin_grad_sync(model)        # last micro-step: finished buckets are reduced during the backward
        if staged:
            model.forward_resident(False)
        else:
            tloss = model(data, False)
        if last_micro:
            if comm is not None and not zero1:
                comm.all_reduce_grads(model)
            optimizer.step(lr_factor=scheduler.factor(), clip_max_norm=max_norm, grad_div=float(world))
            scheduler.step()
        if deferred:
            model.push_losses()                # the step's sums stay on the device (no host wait)
            parked += 1
        nxt = next(it, None)
        if nxt is not None and pipelined:
            model.prefetch(nxt)                # host work of the next batch, beside the step the device is still running
        if deferred:
            if parked == model.LOSS_RING or nxt is None:
                for tloss, ws in model.drain_losses():      # (synchronises: once per LOSS_RING steps and at the end of the epoch)
                    for i in range(n_tasks):
                        training_losses[i] += tloss[i] * ws[i]
                        training_weights[i] += ws[i]
                parked = 0
        else:
            if staged:
                tloss = model.losses(False)        # (synchronises)
            for i in range(n_tasks):
                w = model.last_weight_sums[i]
                training_losses[i] += tloss[i] * w
                training_weights[i] += w
        if nxt is not None and staged:
            if pipelined:
                model.swap_batch()
            else:
                model.upload(nxt)
        data = nxt
        step += 1
    return reduce_mean(comm, training_losses, training_weights)


def train_step_unfused(model, optimizer, data, task_weights, masks=None, max_norm=1.0, lr_factor=1.0):
    """One optimizer step spelled exactly like train.py:259-276 (separate clip pass), for parity tests."""
    model.set_loss_weights(task_weights, 1)
    tloss = model(data, False, masks=masks)
    norm = clip_grad_norm_(model, max_norm)
    optimizer.step(lr_factor=lr_factor)
    return tloss, norm


def checkpoint_model(datadir, model, optimizer, scheduler, config, epoch, training_loss, test_loss, task_weights, save,
                     basename="transformer.masked", gather=None, write=True):
    """train.py:431-497 (rank-local-0 only is the caller's job).  The checkpoint is an `.npz` with the reference's
    state-dict key names + AdamW moments + scheduler state + config JSON (a torch-pickle converter is SURVEY N3);
    the metrics CSV has the reference's exact header and row format (train.py:483-494)."""
    import json
    import os
    # (row-sharded table: EVERY rank calls this with `gather` = the ranks' HostGroup -- the table rows and their moments are
    # collected over the control plane -- and `write` = whether this rank writes the files)
    if save:
        gk = {"gather": gather} if gather is not None else {}
        blob = {"model/" + k: v for k, v in model.state_dict(include_frozen=False, **gk).items() if not k.startswith("watch_head.")}
        if optimizer is not None:
            osd = optimizer.state_dict(**gk)
            blob["optimizer/step"] = np.array([osd["step"]])
            for n, s in osd["state"].items():
                blob["optimizer/exp_avg/" + n] = s["exp_avg"]
                blob["optimizer/exp_avg_sq/" + n] = s["exp_avg_sq"]
            blob["optimizer/lr"] = np.array([float(osd.get("lr", config.get("learning_rate", 1e-4)))])
        if scheduler is not None:
            blob["scheduler/last_epoch"] = np.array([scheduler.state_dict()["last_epoch"]])
            fn = getattr(scheduler, "fn", None)      # the schedule's own parameters: npz2pt rebuilds torch's LambdaLR state from them
            lam = fn.reference_state() if hasattr(fn, "reference_state") else {"steps": int(getattr(fn, "steps", 0))}
            blob["scheduler/lambda"] = np.frombuffer(json.dumps(lam).encode(), np.uint8)
        blob["config"] = np.frombuffer(json.dumps(config).encode(), np.uint8)
        blob["epoch"] = np.array([epoch])
        blob["training_loss"] = np.array(training_loss, np.float64)
        blob["test_loss"] = np.array(test_loss, np.float64)
        if write:
            np.savez(os.path.join(datadir, basename + ".npz"), **blob)
    if not write:
        return
    names = [f"{m}.{metric}" for m in ALL_MEDIUMS for metric in ALL_METRICS]
    csv_fn = os.path.join(datadir, basename + ".csv")
    if epoch < 0:
        with open(csv_fn, "w") as f:
            f.write(",".join(["epoch", "training_loss", "test_loss"] + names) + "\n")
    with open(csv_fn, "a") as f:
        vals = [epoch, wsum(training_loss, task_weights), wsum(test_loss, task_weights)] + list(test_loss)
        f.write(",".join([str(x) for x in vals]) + "\n")


def load_checkpoint(path, model, optimizer=None, scheduler=None):
    """Resume (train.py:657-665, 690-700): returns (epoch, config)."""
    import json
    z = np.load(path)
    sd = {k[len("model/"):]: z[k] for k in z.files if k.startswith("model/")}
    model.load_state_dict(sd, strict=False)
    if optimizer is not None and "optimizer/step" in z.files:
        state = {}
        for k in z.files:
            if k.startswith("optimizer/exp_avg/"):
                n = k[len("optimizer/exp_avg/"):]
                state[n] = {"exp_avg": z[k], "exp_avg_sq": z["optimizer/exp_avg_sq/" + n]}
        optimizer.load_state_dict({"step": int(z["optimizer/step"][0]), "state": state})
    if scheduler is not None and "scheduler/last_epoch" in z.files:
        scheduler.load_state_dict({"last_epoch": int(z["scheduler/last_epoch"][0])})
    return int(z["epoch"][0]), json.loads(bytes(z["config"]).decode())


def get_run_config(finetune):
    """train.py:591-604: epochs / global batch / local batch / gradient accumulation of the two modes."""
    if finetune:
        num_epochs, global_batch, local_batch = 16, 32, 16
    else:
        num_epochs, global_batch, local_batch = 64, 512, 64
    assert global_batch % local_batch == 0
    return {"num_epochs": num_epochs, "global_batch_size": global_batch, "local_batch_size": local_batch}


def train(model, optimizer, scheduler, dataloaders, config, datadir, task_weights, num_epochs, grad_accum_steps,
          comm=None, rank=0, starting_epoch=0, basename="transformer.masked", log=print, gather=None):
    """The epoch loop of train() (train.py:697-757): initial evaluation (CSV row of epoch start-1), then per epoch
    train_epoch -> evaluate_metrics -> early stopper -> checkpoint when the stopper says the model improved; stops early
    when the stopper runs out of patience.  Returns the list of (epoch, training_loss, test_loss)."""
    stopper = make_early_stopper(config)
    get_loss = lambda: evaluate_metrics(model, dataloaders["test"], comm)
    # DDP broadcasts rank 0's parameters when it wraps the model (train.py:678-682); every rank here built its own from the same seed or
    # the same checkpoint, and the ranks check that they agree: now (after init / resume) and at the end of every epoch
    from .dist import assert_replicas_equal
    consistent = lambda when: assert_replicas_equal(model, comm, when) if hasattr(model, "param_checksum") else None
    consistent(f"before epoch {starting_epoch}")
    initial_loss = get_loss()
    log(f"Initial Loss: {wsum(initial_loss, task_weights)}, {initial_loss}")
    stopper(wsum(initial_loss, task_weights))
    if rank == 0 or gather is not None:
        checkpoint_model(datadir, model, optimizer, scheduler, config, starting_epoch - 1, initial_loss, initial_loss,
                         task_weights, bool(config.get("finetune")), basename, gather=gather, write=rank == 0)
    history = []
    for epoch in range(starting_epoch, num_epochs):
        training_loss = train_epoch(model, dataloaders["training"], optimizer, scheduler, task_weights, grad_accum_steps, comm)
        log(f"Epoch: {epoch}, Training Loss: {wsum(training_loss, task_weights)} {training_loss}, LR factor: {scheduler.factor()}")
        consistent(f"after epoch {epoch}")
        test_loss = get_loss()
        log(f"Epoch: {epoch}, Test Loss: {wsum(test_loss, task_weights)} {test_loss}")
        stopper(wsum(test_loss, task_weights))
        if rank == 0 or gather is not None:
            checkpoint_model(datadir, model, optimizer, scheduler, config, epoch, training_loss, test_loss, task_weights,
                             stopper.save_model, basename, gather=gather, write=rank == 0)
        history.append((epoch, training_loss, test_loss))
        if stopper.early_stop:
            break
    return history
