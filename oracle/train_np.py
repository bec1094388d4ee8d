"""CPU oracle for the training-loop arithmetic around the model (numpy / pure Python).

TEST INFRASTRUCTURE ONLY (see oracle/model_np.py header).  `train.py` below =
/root/reference/notebooks/Training/transformer.py.  Pinned by
tests/test_oracle_golden.py against fixtures generated from the reference's own
functions (tests/golden/host_fns.npz, model_*.npz opt/*).
"""
import numpy as np

from . import model_np, synth


def wsd_factor(step, warmup_steps, total_steps, decay_ratio=0.1, final_ratio=0.1):
    """train.py:310-328 WSDScheduler.__call__ as the lower envelope of two ramps: the warm-up line through the origin capped
    at 1, and the line that leaves 1 at the start of the decay and reaches `final_ratio` at `total_steps`."""
    n_decay = int(total_steps * decay_ratio)
    decay_start = total_steps - n_decay
    assert decay_start >= warmup_steps
    s = min(max(int(step), 0), total_steps)
    rising = min(1.0, s / max(1, warmup_steps))
    falling = 1.0 - (1.0 - final_ratio) * max(0, s - decay_start) / max(1, n_decay)
    return min(rising, falling)


_LOSS_SCALE = np.array([4.618602403897067, 1.1958987168236102, 2.5443243303769867, 1.0527565486045412])   # train.py:395-400, TASKS order


def make_task_weights(finetune_medium=None, finetune_metric=None):
    """train.py:379-406: (medium weight x metric weight), normalised to sum 1, divided by each task's loss scale."""
    per_medium = np.array([0.25, 1.0]) if finetune_medium is None else np.eye(2)[finetune_medium]
    per_metric = np.array([1.0, 0.25]) if finetune_metric is None else np.eye(2)[model_np.ALL_METRICS.index(finetune_metric)]
    w = np.array([per_medium[m] * per_metric[model_np.ALL_METRICS.index(k)] for (m, k) in model_np.TASKS])
    return list(w / w.sum() / _LOSS_SCALE)


def minimize_quadratic(x, y):
    """train.py:187-196: the value at the vertex of the parabola through three points (a constant when the points are level)."""
    y = np.asarray(y, np.float64)
    if np.ptp(y) == 0:
        return float(y[0])
    a, b, c = np.polyfit(np.asarray(x, np.float64), y, 2)
    return float(c - b * b / (4 * a))


def block_permutation_indices(userid, block_perm):
    """train.py:53-68 get_index_permutation with the block permutation supplied: cut the stream where the userid changes
    and emit the users' index runs in `block_perm` order."""
    userid = np.asarray(userid)
    runs = np.split(np.arange(len(userid), dtype=np.int64), np.flatnonzero(np.diff(userid)) + 1)
    return np.concatenate([runs[b] for b in block_perm])


class EarlyStopper:
    """train.py:350-372: patience counted against the best score seen by a relative margin; `save_model` whenever the score
    beats the best SAVED score (any margin)."""

    def __init__(self, patience, rtol):
        self.patience, self.rtol = patience, rtol
        self.counter = 0
        self.early_stop = self.save_model = False
        self._bar = self._saved = float("inf")

    def __call__(self, score):
        improved = score < self._bar * (1 - self.rtol)
        self.counter = 0 if improved else self.counter + 1
        if improved:
            self._bar = score
        self.early_stop = self.early_stop or self.counter >= self.patience
        self.save_model = score < self._saved
        self._saved = min(self._saved, score)


def clip_grad_norm(G, max_norm=1.0):
    """train.py:273 torch.nn.utils.clip_grad_norm_: g *= min(1, max_norm/(norm+1e-6))."""
    total = np.sqrt(sum(float((np.asarray(g, np.float64) ** 2).sum()) for g in G.values()))
    coef = min(1.0, max_norm / (total + 1e-6))
    return {k: g * coef for k, g in G.items()}, total


class AdamW:
    """train.py:285-298 create_optimizer -> torch AdamW semantics: decoupled decay
    0.1 for tensors with dim>=2 (incl. embedding tables), 0 otherwise; betas
    (0.9,0.95); eps 1e-8; bias correction."""

    def __init__(self, P, names, lr, betas=(0.9, 0.95), eps=1e-8, wd=0.1):
        self.lr, self.b1, self.b2, self.eps, self.wd = lr, betas[0], betas[1], eps, wd
        self.names = list(names)
        self.m = {k: np.zeros_like(P[k]) for k in self.names}
        self.v = {k: np.zeros_like(P[k]) for k in self.names}
        self.t = 0

    def step(self, P, G, lr_factor=1.0):
        self.t += 1
        lr = self.lr * lr_factor
        bc1 = 1 - self.b1 ** self.t; bc2 = 1 - self.b2 ** self.t
        for k in self.names:
            g = G[k]
            wd = self.wd if P[k].ndim >= 2 else 0.0
            P[k] = P[k] * (1 - lr * wd)
            self.m[k] = self.b1 * self.m[k] + (1 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1 - self.b2) * g * g
            denom = np.sqrt(self.v[k]) / np.sqrt(bc2) + self.eps
            P[k] = P[k] - (lr / bc1) * (self.m[k] / denom)
        return P


def train_step(cfg, P, opt, d_flat, watch_mask, rating_mask, task_w, lr_factor=1.0, dtype=np.float64):
    """One optimizer step of train_epoch (train.py:256-276), grad_accum=1."""
    model = model_np.OracleModel(cfg, P, dtype)
    dm = model_np.mask_tokens(cfg, model_np.reshape_batch(cfg, d_flat), watch_mask, rating_mask)
    losses, G = model.forward(dm, False, True, task_w)
    names = synth.trainable_names(cfg)
    G = {k: G[k] for k in names}
    G, norm = clip_grad_norm(G, 1.0)
    P2 = dict(model.P)
    opt.step(P2, G, lr_factor)
    return P2, losses, norm
