"""ctypes binding of oracle/cpu_step.cpp (the C++ / OpenMP restatement of one training step).

TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/model_np.py header).  Parameter / gradient pointer tables are
built from the reference's state-dict names (transformer.model.py:346-359) in the order of the P_* / L_* enums of cpu_step.cpp.
"""
import ctypes
import os
import subprocess

import numpy as np

from . import model_np, synth

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcpu_step.so")
_lib = None


class Cfg(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in ("L", "H", "KV", "D", "I", "S", "V0", "V1", "M", "K", "vs_status", "vs_gender", "vs_source", "rows")] + \
               [("min_ts", ctypes.c_double), ("max_ts", ctypes.c_double), ("rating_mean", ctypes.c_float), ("rating_std", ctypes.c_float)]


class Batch(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("userid", "tmid", "gender", "source", "matchedid", "status", "time", "rating", "progress")] + \
               [("label", ctypes.c_void_p * 4), ("weight", ctypes.c_void_p * 4), ("position", ctypes.c_void_p * 4)]


def build():
    subprocess.run(["make", "-C", HERE, "libcpu_step.so"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = ctypes.CDLL(LIB_PATH)
        L.cpu_step_threads.restype = ctypes.c_int
        L.cpu_step_forward_backward.restype = ctypes.c_int
        L.cpu_step_forward_backward.argtypes = [ctypes.POINTER(Cfg), ctypes.c_void_p, ctypes.POINTER(Batch), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.cpu_step_clip_adamw.restype = ctypes.c_double
        L.cpu_step_clip_adamw.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_float] * 4 + [ctypes.c_int, ctypes.c_float]
        L.cpu_step_sgemm_nt.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
        L.cpu_step_isa.restype = ctypes.c_int
        L.cpu_step_set_threads.argtypes = [ctypes.c_int]
        L.cpu_step_set_operand_round.argtypes = [ctypes.c_int]
        # one thread per CPU this process is GRANTED: OpenMP's default is one per visible core, and 256 spinning threads on a
        # 16-CPU cgroup quota run the step ~10x slower (measured on the GPU box)
        L.cpu_step_set_threads(host_cpus())
        _lib = L
    return _lib


def host_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup's CPU quota (a container on a big host sees
    every core in `nproc` but is throttled to its share; one thread per allowed CPU, not per visible one)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def param_order(cfg):
    """state-dict names in the order of cpu_step.cpp's pointer table"""
    names = ["item_embedding.matchedid_embedding.embedding.weight", "item_embedding.metadata_embedding.embedding.weight",
             "item_embedding.projection_layer.weight", "item_embedding.projection_layer.bias",
             "action_embedding.periodic_time_cos", "action_embedding.periodic_time_sin",
             "action_embedding.status_embedding.embedding.weight", "action_embedding.gender_embedding.embedding.weight",
             "action_embedding.source_embedding.embedding.weight", "action_embedding.linear.weight", "action_embedding.linear.bias",
             "transformers.norm.scale", "rating_head.0.weight", "rating_head.0.bias", "rating_head.2.weight", "rating_head.2.bias"]
    for l in range(cfg["num_layers"]):
        p = f"transformers.layers.{l}."
        names += [p + s for s in ("attn.q_proj.weight", "attn.k_proj.weight", "attn.v_proj.weight", "attn.output_proj.weight",
                                  "mlp.w1.weight", "mlp.w2.weight", "mlp.w3.weight", "sa_norm.scale", "mlp_norm.scale")]
    return names


def _table(arrays):
    t = (ctypes.c_void_p * len(arrays))()
    for i, a in enumerate(arrays):
        t[i] = a.ctypes.data
    return t


def release():
    """free the work buffers the library keeps between steps"""
    if _lib is not None:
        _lib.cpu_step_release()


class CpuStep:
    """One model instance: fp32 parameters (copied), gradients, AdamW moments; `step` = forward + backward + clip + AdamW."""

    def __init__(self, cfg, P, lr=1e-4, operand_round=None):
        """operand_round="bf16": the rounding points of model_np.OracleModel(operand_round="bf16") (every array the HIP path stores
        as a bf16 GEMM operand is rounded where the kernels store it; float accumulation).  The metadata table must then hold
        bfloat16-representable values (it is not copied)."""
        assert not cfg.get("finetune")
        assert operand_round in (None, "bf16")
        self.operand_round = operand_round
        self.cfg = cfg
        self.names = param_order(cfg)
        assert set(self.names) == set(synth.param_shapes(cfg)), "parameter table does not cover the state dict"
        # trainable tensors are copied (AdamW updates them in place); the frozen metadata table (4.9 GB at cfg-3) is only read
        self.P = {k: np.ascontiguousarray(P[k], np.float32) if k in synth.FROZEN else np.ascontiguousarray(P[k], np.float32).copy()
                  for k in self.names}
        self.G = {k: np.zeros_like(self.P[k]) for k in self.names if k not in synth.FROZEN}
        self.train_names = [k for k in self.names if k not in synth.FROZEN]
        self.m = {k: np.zeros_like(self.P[k]) for k in self.train_names}
        self.v = {k: np.zeros_like(self.P[k]) for k in self.train_names}
        self.lr, self.t = lr, 0
        vs = cfg["vocab_sizes"]
        self.c = Cfg(cfg["num_layers"], cfg["num_heads"], cfg["num_kv_heads"], cfg["embed_dim"], cfg["intermediate_dim"],
                     cfg["max_sequence_length"], vs["0_matchedid"], vs["1_matchedid"], cfg["metadata_emb_size"], cfg["mask_topk"],
                     vs["status"], vs["gender"], vs["source"], 0, cfg["min_ts"], cfg["max_ts"], cfg["rating_mean"], cfg["rating_std"])

    def forward_backward(self, dm, task_w):
        """dm: already-masked batch of (rows, S) arrays (model_np.mask_tokens).  Returns (4 losses, {name: gradient})."""
        rows = dm["userid"].shape[0]
        self.c.rows = rows
        keep = []

        def arr(k, dt):
            a = np.ascontiguousarray(dm[k], dt); keep.append(a); return a.ctypes.data
        b = Batch(arr("userid", np.int32), arr("token_mask_ids", np.int32), arr("gender", np.int32), arr("source", np.int32),
                  arr("matchedid", np.int32), arr("status", np.int32), arr("time", np.float64), arr("rating", np.float32),
                  arr("progress", np.float32))
        for ti, (medium, metric) in enumerate(model_np.TASKS):
            b.label[ti] = arr(f"{medium}.{metric}.label", np.float32)
            b.weight[ti] = arr(f"{medium}.{metric}.weight", np.float32)
            b.position[ti] = arr(f"{medium}.{metric}.position", np.int32)
        pt = _table([self.P[k] for k in self.names])
        dummy = np.zeros(1, np.float32)
        gt = _table([self.G.get(k, dummy) for k in self.names])
        tw = np.ascontiguousarray(task_w, np.float32)
        losses = np.zeros(4, np.float32)
        lib().cpu_step_set_operand_round(1 if self.operand_round == "bf16" else 0)
        if self.operand_round == "bf16":
            meta = self.P["item_embedding.metadata_embedding.embedding.weight"]
            probe = meta[:: max(1, meta.shape[0] // 64)]
            assert np.array_equal(model_np.bf16_round(probe), probe), "operand_round: the metadata table must be bfloat16-representable"
        try:
            rc = lib().cpu_step_forward_backward(ctypes.byref(self.c), pt, ctypes.byref(b), tw.ctypes.data, losses.ctypes.data, gt)
        finally:
            lib().cpu_step_set_operand_round(0)
        if rc != 0:
            raise ValueError("cpu_step: index out of range in the batch")
        return [float(x) for x in losses], self.G

    def clip_adamw(self, lr_factor=1.0, betas=(0.9, 0.95), eps=1e-8, wd=0.1):
        """train.py:273 + :285-298; returns the gradient norm before clipping."""
        self.t += 1
        n = len(self.train_names)
        numel = np.array([self.P[k].size for k in self.train_names], np.int64)
        decay = np.array([wd if self.P[k].ndim >= 2 else 0.0 for k in self.train_names], np.float32)
        return lib().cpu_step_clip_adamw(n, _table([self.P[k] for k in self.train_names]), _table([self.G[k] for k in self.train_names]),
                                         _table([self.m[k] for k in self.train_names]), _table([self.v[k] for k in self.train_names]),
                                         numel.ctypes.data, decay.ctypes.data, self.lr * lr_factor, betas[0], betas[1], eps, self.t, 1.0)


def sgemm_nt(A, B):
    A = np.ascontiguousarray(A, np.float32); B = np.ascontiguousarray(B, np.float32)
    C = np.empty((A.shape[0], B.shape[0]), np.float32)
    lib().cpu_step_sgemm_nt(A.shape[0], B.shape[0], A.shape[1], A.ctypes.data, B.ctypes.data, C.ctypes.data)
    return C
