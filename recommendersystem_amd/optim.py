"""create_optimizer(model, config) -- notebooks/Training/transformer.py:285-298.

AdamW over the flat parameter buffer: decoupled weight decay 0.1 on tensors with
dim >= 2 (incl. the embedding tables), 0 on biases / norm scales / phases, betas
(0.9, 0.95), eps 1e-8.  `step()` also performs optimizer.zero_grad()
(train.py:274-275) and can fuse clip_grad_norm_ (train.py:273) and the
data-parallel gradient mean into the single pass over the parameters.
"""
import ctypes as C

import numpy as np

from ._lib import check, lib


class AdamW:
    def __init__(self, model, lr, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1):
        self.model = model
        self.lr = lr
        self._h = C.c_void_p()
        check(lib().rsys_adamw_create(model._h, lr, betas[0], betas[1], eps, weight_decay, C.byref(self._h)))

        self._zero1 = None

    def enable_zero1(self, comm):
        """(beyond the reference, opt-in) ZeRO-1: keep AdamW moments for this rank's 1/world of the parameters only.  step() then
        reduce-scatters the gradient itself (do NOT call comm.all_reduce_grads / begin_grad_sync), updates the rank's part and
        gathers the parameters.  Call before the first step; replicated pretraining model only."""
        check(lib().rsys_adamw_set_zero1(self._h, comm.rank, comm.world))
        self._zero1 = comm

    def step(self, lr_factor=1.0, clip_max_norm=0.0, grad_div=1.0):
        if self._zero1 is not None:
            check(lib().rsys_adamw_step_zero1(self._h, self._zero1._h, lr_factor, clip_max_norm, grad_div))
        else:
            check(lib().rsys_adamw_step(self._h, lr_factor, clip_max_norm, grad_div))

    def zero_grad(self, set_to_none=True):
        self.model.zero_grad()

    def state_dict(self, gather=None):
        """AdamW moments by parameter name.  `gather` (the ranks' HostGroup): moments of a row-sharded table whole, and -- ZeRO-1 --
        the moments every rank keeps for its 1/world of the flat parameter range summed into whole tensors (every rank calls this;
        the result is the replicated optimizer's state_dict, so a checkpoint does not depend on the world size it was written at)."""
        from .model import gather_rows
        zero1_parts = self._zero1 is not None and self._zero1.world > 1
        if zero1_parts and gather is None:
            raise ValueError("ZeRO-1 optimizer state is partitioned over the ranks: call state_dict(gather=<the ranks' dist.HostGroup>) on "
                             "every rank (train.train passes its `gather` argument through)")
        st = C.c_int32()
        check(lib().rsys_adamw_state_get(self._h, None, None, None, 0, C.byref(st)))
        state = {}
        sharded = bool(self.model.config.get("table_shard"))
        for n, shape, tr in self.model.named_parameters():
            if not tr:
                continue
            m = np.empty(shape, np.float32); v = np.empty(shape, np.float32)
            check(lib().rsys_adamw_state_get(self._h, n.encode(), m.ctypes.data, v.ctypes.data, m.size, None))
            if zero1_parts:      # the ranks' parts are disjoint (zeros elsewhere): their sum is the tensor
                mv = np.stack([m, v])
                parts = gather.all_gather_bytes(mv.tobytes())
                mv = np.sum([np.frombuffer(p, np.float32).reshape(mv.shape) for p in parts], axis=0, dtype=np.float32)
                m, v = np.ascontiguousarray(mv[0]), np.ascontiguousarray(mv[1])
            if gather is not None and sharded and n in self.model.TABLE_KEYS:
                m, v = gather_rows(gather, m), gather_rows(gather, v)
            state[n] = {"exp_avg": m, "exp_avg_sq": v}
        return {"step": st.value, "lr": self.lr, "state": state}

    def load_state_dict(self, sd):
        """Whole tensors in; a ZeRO-1 optimizer keeps the part of each that falls into this rank's range (enable_zero1 first)."""
        check(lib().rsys_adamw_state_set(self._h, None, None, None, 0, int(sd["step"])))
        lo, hi = self.model.table_rows()
        for n, s in sd["state"].items():
            m, v = np.asarray(s["exp_avg"]), np.asarray(s["exp_avg_sq"])
            if n in self.model.TABLE_KEYS and m.shape[0] != hi - lo:      # whole-table moments: keep this rank's rows
                m, v = m[lo:hi], v[lo:hi]
            m = np.ascontiguousarray(m, np.float32); v = np.ascontiguousarray(v, np.float32)
            check(lib().rsys_adamw_state_set(self._h, n.encode(), m.ctypes.data, v.ctypes.data, m.size, -1))

    def close(self):
        if self._h:
            lib().rsys_adamw_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def create_optimizer(model, config):
    return AdamW(model, lr=config["learning_rate"], betas=(0.9, 0.95), weight_decay=0.1)


def clip_grad_norm_(model, max_norm):
    """torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm), train.py:273 -> total norm."""
    out = C.c_float()
    check(lib().rsys_clip_grad_norm(model._h, max_norm, C.byref(out)))
    return out.value
