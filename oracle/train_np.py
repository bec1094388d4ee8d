"""CPU oracle for the training-loop arithmetic around the model (numpy / pure Python).

TEST INFRASTRUCTURE ONLY (see oracle/model_np.py header).  `train.py` below =
/root/reference/notebooks/Training/transformer.py.  Pinned by
tests/test_oracle_golden.py against fixtures generated from the reference's own
functions (tests/golden/host_fns.npz, model_*.npz opt/*).
"""
import numpy as np

from . import model_np, synth


def wsd_factor(step, warmup_steps, total_steps, decay_ratio=0.1, final_ratio=0.1):
    """train.py:310-328 WSDScheduler.__call__."""
    decay_steps = int(total_steps * decay_ratio)
    stable_steps = total_steps - warmup_steps - decay_steps
    assert stable_steps >= 0
    s = max(0, min(int(step), total_steps))
    if s <= warmup_steps:
        return s / max(1, warmup_steps)
    if s <= warmup_steps + stable_steps:
        return 1.0
    prog = (s - warmup_steps - stable_steps) / max(1, decay_steps)
    return 1.0 - (1.0 - final_ratio) * prog


def make_task_weights(finetune_medium=None, finetune_metric=None):
    """train.py:379-406."""
    scale = {(0, "watch"): 4.618602403897067, (0, "rating"): 1.1958987168236102,
             (1, "watch"): 2.5443243303769867, (1, "rating"): 1.0527565486045412}
    if finetune_metric is None:
        mw = {"watch": 1, "rating": 0.25}
    else:
        mw = {"watch": 0, "rating": 0}; mw[finetune_metric] = 1
    if finetune_medium is None:
        dw = {0: 0.25, 1: 1}
    else:
        dw = {finetune_medium: 1, 1 - finetune_medium: 0}
    w = [dw[m] * mw[k] for (m, k) in model_np.TASKS]
    tot = sum(w)
    return [x / tot / scale[t] for x, t in zip(w, model_np.TASKS)]


def minimize_quadratic(x, y):
    """train.py:187-196."""
    if max(y) == min(y):
        return float(max(y))
    A = np.array([[xi ** 2, xi, 1] for xi in x], np.float64)
    a, b, c = np.linalg.solve(A, np.array(y, np.float64))
    xe = -b / (2 * a)
    return float(a * xe ** 2 + b * xe + c)


def block_permutation_indices(userid, block_perm):
    """train.py:53-68 get_index_permutation with the block permutation supplied:
    split at userid change points, emit whole user blocks in `block_perm` order."""
    userid = np.asarray(userid)
    change = np.where(userid[:-1] != userid[1:])[0] + 1
    starts = np.concatenate([[0], change]); ends = np.concatenate([change, [len(userid)]])
    return np.concatenate([np.arange(starts[b], ends[b]) for b in block_perm]).astype(np.int64)


class EarlyStopper:
    """train.py:350-372."""

    def __init__(self, patience, rtol):
        self.patience, self.rtol = patience, rtol
        self.counter = 0
        self.stop_score = float("inf"); self.saved_score = float("inf")
        self.early_stop = False; self.save_model = False

    def __call__(self, score):
        if score < self.stop_score * (1 - self.rtol):
            self.counter = 0; self.stop_score = score
        else:
            self.counter += 1
            if self.counter >= self.patience:
                self.early_stop = True
        if score < self.saved_score:
            self.saved_score = score; self.save_model = True
        else:
            self.save_model = False


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
