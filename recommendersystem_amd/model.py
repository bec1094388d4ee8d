"""Host-side mirror of the reference's model API over the C ABI.

Mirrors notebooks/Training/transformer.model.py ("model.py") as used by
notebooks/Training/transformer.py ("train.py"):

  RecommenderModel(config)                 model.py:346-377   -> rsys_model_create
  model.load_pretrained_embeddings(...)    model.py:379-389   -> rsys_model_load_metadata
  model(d, evaluate) -> 4 losses           model.py:493-529   -> rsys_batch_upload + rsys_forward_backward
  model(d, task)  (inference)              model.py:531-538   -> rsys_infer
  state_dict()/load_state_dict()           train.py:458,664   -> rsys_param_get / rsys_param_set

Difference forced by fusing forward and backward in one device pass: the
reference computes `loss = sum(tloss[i]*task_weights[i])/grad_accum` on the host
and calls loss.backward() (train.py:264-272); here the task weights and the
1/grad_accum scale are given BEFORE the call (`set_loss_weights`) and
`model(d, False)` accumulates the gradient of that weighted sum.
All arrays are numpy; device memory is owned by the library.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import check, lib

ALL_MEDIUMS = [0, 1]
ALL_METRICS = ["watch", "rating"]
_METRICS3 = ["watch", "rating", "status"]
DTYPES = {"fp32": 0, "float32": 0, "f32": 0, "bf16": 1, "bfloat16": 1, "fp8": 2, "float8": 2}   # fp8: bf16 + the float8 trunk linears (transformer.py:671-676)


def _c_config(config, dtype, max_rows):
    vs = config["vocab_sizes"]
    c = _lib.rsys_config()
    c.num_layers = config["num_layers"]; c.num_heads = config["num_heads"]; c.num_kv_heads = config["num_kv_heads"]
    c.embed_dim = config["embed_dim"]; c.intermediate_dim = config["intermediate_dim"]
    c.max_sequence_length = config["max_sequence_length"]
    c.vocab_0 = vs["0_matchedid"]; c.vocab_1 = vs["1_matchedid"]
    c.vocab_status = vs["status"]; c.vocab_gender = vs["gender"]; c.vocab_source = vs["source"]
    c.metadata_dim = config["metadata_emb_size"]
    c.min_ts = float(config["min_ts"]); c.max_ts = float(config["max_ts"])
    c.rating_mean = float(config["rating_mean"]); c.rating_std = float(config["rating_std"])
    c.mask_rate = float(config["mask_rate"]); c.mask_topk = int(config["mask_topk"])
    c.finetune = 1 if config.get("finetune") else 0
    c.finetune_metric = 1 if config.get("finetune_metric") == "rating" else 0
    c.dtype = DTYPES[dtype]
    c.max_rows = int(max_rows)
    c.lora_dropout = float(config.get("lora_dropout", 0.1 if config.get("finetune") else 0.0))   # nn.Dropout(0.1), model.py:238
    shard = config.get("table_shard")            # (rank, world): row-sharded item table (cfg-4); None = replicated
    c.table_shard_rank, c.table_shard_world = (int(shard[0]), int(shard[1])) if shard else (0, 0)
    c.sampled_negatives = int(config.get("sampled_softmax", 0))   # classes sampled per rank and medium (0: full soft-max)
    return c


def precompute_freqs_cis(dim, end, theta=500000.0):
    """model.py:173-179 in float32 (host computes the tables, the device only reads them)."""
    freqs = (1.0 / (np.float32(theta) ** (np.arange(0, dim, 2, dtype=np.float32)[: dim // 2] / np.float32(dim)))).astype(np.float32)
    t = np.arange(end, dtype=np.float32)
    f = np.outer(t, freqs).astype(np.float32)
    return np.cos(f).astype(np.float32), np.sin(f).astype(np.float32)


class RecommenderModel:
    def __init__(self, config, device=0, dtype="bf16", max_rows=None):
        assert config.get("forward", "train") in ("train", "inference")
        self.config = config
        self.device = device
        self.dtype = dtype
        self.max_rows = int(max_rows if max_rows is not None else config.get("local_batch_size", 1))
        self._h = C.c_void_p()
        cc = _c_config(config, dtype, self.max_rows)
        check(lib().rsys_model_create(C.byref(cc), device, C.byref(self._h)))
        hd = config["embed_dim"] // config["num_heads"]
        cos, sin = precompute_freqs_cis(hd, 2 * config["max_sequence_length"])
        check(lib().rsys_model_set_rope(self._h, cos.ctypes.data, sin.ctypes.data, cos.shape[0]))
        self._names = None
        self._task_w = None
        self._grad_scale = 1.0
        self._keep = None
        self.training = True
        self.mask_seed = 0x3A5C
        self._step = 0
        if config.get("deterministic"):
            self.set_deterministic(True)

    # ---- lifetime
    def close(self):
        if self._h:
            lib().rsys_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def train(self):
        self.training = True
        return self

    def eval(self):
        self.training = False
        return self

    # ---- parameters
    def init_weights(self, seed=0x1217):
        """self.apply(init_weights), model.py:5-12,360 (device-side Philox normal)."""
        check(lib().rsys_model_init_random(self._h, seed))

    def load_pretrained_embeddings(self, table):
        """model.py:379-389; `table` is the (V, M) float32 `metadata` array of media_embeddings.h5, or, as in the
        reference, the directory that holds that file."""
        if isinstance(table, (str, os.PathLike)):
            from . import h5
            with h5.File(os.path.join(table, "media_embeddings.h5")) as f:
                table = f["metadata"]
        table = np.ascontiguousarray(table, np.float32)
        check(lib().rsys_model_load_metadata(self._h, table.ctypes.data, table.shape[0], table.shape[1]))

    def random_pretrained_embeddings(self, seed=0x3E7A):
        check(lib().rsys_model_random_metadata(self._h, seed))

    def set_deterministic(self, on=True):
        """bitwise reproducible training steps (fixed summation order everywhere; a few percent slower); also `config["deterministic"]`"""
        check(lib().rsys_model_set_deterministic(self._h, 1 if on else 0))

    def set_split_table_reduce(self, on=True):
        """opt-in (replicated table, bf16): reduce the item table's gradient in two parts -- the heads' part early and out of place
        under the trunk backward, the batch's token rows as a gathered list in the tail (`Comm.begin_grad_sync` arms it per step)"""
        check(lib().rsys_model_set_split_table_reduce(self._h, 1 if on else 0))

    def set_shard_comm(self, comm):
        """row-sharded table mode: the communicator of the row exchange and the vocabulary-parallel cross entropy"""
        check(lib().rsys_model_set_shard_comm(self._h, comm._h if comm is not None else None))

    def table_rows(self):
        """[lo, hi) of the (V + 1)-row item table held by this model"""
        lo = C.c_int64(); hi = C.c_int64()
        check(lib().rsys_table_rows(self._h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def named_parameters(self):
        """[(name, shape, trainable)] in state_dict order."""
        if self._names is None:
            n = C.c_int32()
            check(lib().rsys_param_count(self._h, C.byref(n)))
            out = []
            for i in range(n.value):
                name = C.create_string_buffer(256)
                shape = (C.c_int64 * 2)()
                nd = C.c_int32(); tr = C.c_int32()
                check(lib().rsys_param_info(self._h, i, name, 256, C.byref(shape), C.byref(nd), C.byref(tr)))
                shp = (shape[0],) if nd.value == 1 else (shape[0], shape[1])
                out.append((name.value.decode(), tuple(int(x) for x in shp), bool(tr.value)))
            self._names = out
        return self._names

    def _shape(self, name):
        base = name[len("watch_head."):] if name.startswith("watch_head.") else name
        for n, s, _ in self.named_parameters():
            if n == base:
                return base, s
        raise KeyError(name)

    def get_parameter(self, name):
        base, shape = self._shape(name)
        out = np.empty(shape, np.float32)
        check(lib().rsys_param_get(self._h, base.encode(), out.ctypes.data, out.size))
        return out

    def set_parameter(self, name, value):
        base, shape = self._shape(name)
        v = np.ascontiguousarray(value, np.float32)
        assert v.shape == tuple(shape), (name, v.shape, shape)
        check(lib().rsys_param_set(self._h, base.encode(), v.ctypes.data, v.size))

    def grad(self, name):
        base, shape = self._shape(name)
        out = np.empty(shape, np.float32)
        check(lib().rsys_grad_get(self._h, base.encode(), out.ctypes.data, out.size))
        return out

    TABLE_KEYS = ("item_embedding.matchedid_embedding.embedding.weight", "item_embedding.metadata_embedding.embedding.weight")

    def state_dict(self, include_frozen=True, gather=None):
        """The reference's state dict.  With a row-sharded item table the two table tensors are this rank's rows; pass
        `gather` = the ranks' `dist.HostGroup` to get the full tables on every rank (checkpoints: the reference's layout)."""
        sd = {}
        for n, _, tr in self.named_parameters():
            if not include_frozen and "metadata_embedding" in n:
                continue
            sd[n] = self.get_parameter(n)
            if gather is not None and n in self.TABLE_KEYS and self.config.get("table_shard"):
                sd[n] = gather_rows(gather, sd[n])
        for k in list(sd):
            if k.startswith("item_embedding."):       # watch_head shares item_embedding (model.py:354)
                sd["watch_head." + k] = sd[k]
        return sd

    def load_state_dict(self, sd, strict=True):
        """Table tensors may be given whole ((V + 1) rows, the reference's layout): a row-sharded model keeps its rows."""
        names = [n for n, _, _ in self.named_parameters()]
        assert "item_embedding.fused_embedding" not in sd          # model.py:135-137
        lo, hi = self.table_rows()
        for n in names:
            if n in sd:
                v = np.asarray(sd[n])
                if n in self.TABLE_KEYS and v.shape[0] != hi - lo:
                    v = v[lo:hi]
                self.set_parameter(n, v)
            elif strict:
                raise KeyError(f"missing key {n}")
        if strict:
            extra = [k for k in sd if k not in names and not k.startswith("watch_head.")]
            if extra:
                raise KeyError(f"unexpected keys {extra}")

    def zero_grad(self):
        check(lib().rsys_zero_grad(self._h))

    # ---- forward / backward
    def set_loss_weights(self, task_weights, grad_accum_steps=1):
        """task weights of train.py:264-267 (order ALL_MEDIUMS x ALL_METRICS) and the 1/grad_accum scale."""
        self._task_w = [float(x) for x in task_weights]
        self._grad_scale = 1.0 / float(grad_accum_steps)

    def _c_batch(self, d, masks=None):
        """the rsys_batch record of the 27 flat arrays (+ optional masks / RoPE positions) and the arrays it points into"""
        S = self.config["max_sequence_length"]
        n = int(np.asarray(d["userid"]).size)
        assert n % S == 0, "batch must hold whole rows of max_sequence_length"
        b = _lib.rsys_batch()
        b.rows = n // S
        keep = []

        def arr(x, dt):
            a = np.ascontiguousarray(np.asarray(x).reshape(-1), dt)
            assert a.size == n
            keep.append(a)
            return a.ctypes.data

        b.userid = arr(d["userid"], np.int32); b.token_mask_ids = arr(d["token_mask_ids"], np.int32)
        b.gender = arr(d["gender"], np.int32); b.source = arr(d["source"], np.int32)
        b.matchedid = arr(d["matchedid"], np.int32); b.status = arr(d["status"], np.int32)
        b.time = arr(d["time"], np.float64); b.rating = arr(d["rating"], np.float32); b.progress = arr(d["progress"], np.float32)
        zf = np.zeros(n, np.float32); zi = np.zeros(n, np.int32)
        for m in ALL_MEDIUMS:
            for j, metric in enumerate(_METRICS3):
                k = m * 3 + j
                b.label[k] = arr(d.get(f"{m}.{metric}.label", zf), np.float32)
                b.weight[k] = arr(d.get(f"{m}.{metric}.weight", zf), np.float32)
                b.position[k] = arr(d.get(f"{m}.{metric}.position", zi), np.int32)
        if masks is not None:
            b.watch_mask = arr(masks[0], np.uint8); b.rating_mask = arr(masks[1], np.uint8)
        if "rope_input_pos" in d:
            b.rope_input_pos = arr(d["rope_input_pos"], np.int32)
        return b, keep

    def upload(self, d, masks=None):
        """to_device, train.py:178-184.  d: the 27 flat arrays (any shape with rows*S elements)."""
        b, keep = self._c_batch(d, masks)
        check(lib().rsys_batch_upload(self._h, C.byref(b)))
        self._keep = keep

    @property
    def can_prefetch(self):
        """rsys_batch_prefetch / rsys_batch_swap: the next batch staged and copied beside the running step (replicated table)"""
        return self.config.get("table_shard") is None and not getattr(self, "_no_prefetch", False)   # (_no_prefetch: A/B switch of the tests)

    def prefetch(self, d, masks=None):
        """Check, pack and copy the NEXT batch while the step already enqueued still runs on the device (the reference's DataLoader
        workers + non_blocking to_device, train.py:162-165,178-184); `swap_batch` makes it the resident batch."""
        b, keep = self._c_batch(d, masks)
        check(lib().rsys_batch_prefetch(self._h, C.byref(b)))

    def swap_batch(self):
        check(lib().rsys_batch_swap(self._h))

    def forward_resident(self, evaluate, step=None):
        """One pass over the batch already resident on the device (asynchronous)."""
        if step is None:
            step = self._step
            self._step += 1
        if evaluate:
            tw = None
        else:
            assert self._task_w is not None, "call set_loss_weights(task_weights, grad_accum_steps) first"
            tw = (C.c_float * 4)(*self._task_w)
        check(lib().rsys_forward_backward(self._h, 1 if evaluate else 0, tw, self._grad_scale, self.mask_seed, step))

    def losses(self, evaluate=False):
        """Synchronises; returns the reference's loss list (rating entries are 3-lists when evaluate)."""
        lo = (C.c_float * 12)(); ws = (C.c_float * 4)()
        check(lib().rsys_losses_get(self._h, C.byref(lo), C.byref(ws)))
        out = []
        for ti in range(4):
            if evaluate and ti % 2 == 1:
                out.append([float(lo[3 * ti + k]) for k in range(3)])
            else:
                out.append(float(lo[3 * ti]))
        self.last_weight_sums = [float(x) for x in ws]
        return out

    LOSS_RING = 1024

    def push_losses(self):
        """park the enqueued step's losses and weight sums on the device (no host wait); read them with drain_losses()"""
        check(lib().rsys_losses_push(self._h))

    def drain_losses(self):
        """Synchronises once; [(losses, weight_sums)] of every step parked since the last drain, in order, as losses(False) /
        last_weight_sums would have given them step by step."""
        lo = np.empty((self.LOSS_RING, 12), np.float32); ws = np.empty((self.LOSS_RING, 4), np.float32); n = C.c_int32()
        check(lib().rsys_losses_drain(self._h, lo.ctypes.data, ws.ctypes.data, self.LOSS_RING, C.byref(n)))
        out = [([float(lo[s, 3 * ti]) for ti in range(4)], [float(x) for x in ws[s]]) for s in range(n.value)]
        if out:
            self.last_weight_sums = out[-1][1]
        return out

    def __call__(self, d, evaluate_or_task, masks=None):
        if isinstance(evaluate_or_task, str):
            return self.inference_forward(d, evaluate_or_task)
        evaluate = bool(evaluate_or_task)
        self.upload(d, masks)
        self.forward_resident(evaluate)
        return self.losses(evaluate)

    def inference_forward(self, d, task):
        """model.py:531-538: "retrieval" -> (rows, 2S, D), "ranking" -> (rows, 2S, 1)."""
        self.upload(d)
        S = self.config["max_sequence_length"]; D = self.config["embed_dim"]
        rows = int(np.asarray(d["userid"]).size) // S
        if task == "retrieval":
            out = np.empty((rows, 2 * S, D), np.float32)
            check(lib().rsys_infer(self._h, 0, out.ctypes.data, out.size))
            return out
        if task == "ranking":
            out = np.empty((rows, 2 * S, 1), np.float32)
            check(lib().rsys_infer(self._h, 1, out.ctypes.data, out.size))
            return out
        raise AssertionError(task)

    def inference_select(self, d, task, token_index):
        """The inference forward reporting only the tokens a server reads (embed.py:147-161): `token_index` = flat token indices
        in [0, rows * 2S).  "retrieval" -> (n, D) trunk outputs, "ranking" -> (n,) rating-head values (computed on those rows only)."""
        self.upload(d)
        idx = np.ascontiguousarray(np.asarray(token_index).reshape(-1), np.int32)
        D = self.config["embed_dim"]
        t = {"retrieval": 0, "ranking": 1}[task]
        out = np.empty((idx.size, D) if t == 0 else (idx.size,), np.float32)
        check(lib().rsys_infer_select(self._h, t, idx.ctypes.data, idx.size, out.ctypes.data, out.size))
        return out

    def item_embeddings(self):
        """`model.item_embedding(torch.arange(0, n_0 + n_1))` (register.py:27-29): the (V, D) table E + Wp.Meta + bp."""
        V = self.config["vocab_sizes"]["0_matchedid"] + self.config["vocab_sizes"]["1_matchedid"]
        out = np.empty((V, self.config["embed_dim"]), np.float32)
        check(lib().rsys_item_table(self._h, out.ctypes.data, out.size))
        return out

    def trunk_output(self, rows):
        S = self.config["max_sequence_length"]; D = self.config["embed_dim"]
        out = np.empty((rows, 2 * S, D), np.float32)
        check(lib().rsys_trunk_output_get(self._h, out.ctypes.data, out.size))
        return out

    def debug_get(self, key, rows):
        """Bit-exact read-back of an index-path array of the last forward (rsys_debug_get; parity tests)."""
        S = self.config["max_sequence_length"]; D = self.config["embed_dim"]; n = rows * S
        V = self.config["vocab_sizes"]["0_matchedid"] + self.config["vocab_sizes"]["1_matchedid"]
        if key == "npos":
            out = np.empty(4, np.int32)
        elif key in ("top.n", "top.cap"):
            out = np.empty(1, np.int32)
        elif key == "host_syncs":
            out = np.empty(2, np.int32)
        elif key == "top.sel":
            out = np.empty(int(self.debug_get("top.cap", rows)[0]), np.int32)
        elif key == "top.slot":
            out = np.empty(2 * n, np.int32)
        elif key.startswith("idx."):
            out = np.empty(self.config["mask_topk"] * rows, np.int32)
        elif key.startswith("tokens."):
            out = np.empty(2 * n, np.int32)
        elif key == "embed.x0":
            out = np.empty((2 * n, D), np.float32)
        elif key.startswith("act."):     # act.<layer>.<x|xn|qkv|O|h|hn|ab|g>: saved activations (bf16 ones come back widened to float32)
            f = key.split(".")[2]
            H, KV = self.config["num_heads"], self.config["num_kv_heads"]
            Ip = (self.config["intermediate_dim"] + 127) // 128 * 128 if self.dtype in ("fp8", "float8") else (self.config["intermediate_dim"] + 15) // 16 * 16
            cols = {"x": D, "h": D, "xn": D, "hn": D, "O": D, "qkv": (H + 2 * KV) * (D // H), "ab": 2 * Ip, "g": Ip}[f]
            wide = f in ("x", "h") or self.dtype in ("fp32", "float32", "f32")
            out = np.empty((2 * n, cols), np.float32 if wide else np.uint16)
            check(lib().rsys_debug_get(self._h, key.encode(), out.ctypes.data, out.nbytes))
            return out if wide else (out.astype(np.uint32) << 16).view(np.float32)
        elif key.startswith("dw.") or key.startswith("f8keep."):   # bf16 operands the backward kept (widened to float32)
            f = key.split(".")[2]
            H, KV = self.config["num_heads"], self.config["num_kv_heads"]
            Ip = (self.config["intermediate_dim"] + 127) // 128 * 128 if self.dtype in ("fp8", "float8") else (self.config["intermediate_dim"] + 15) // 16 * 16
            cols = {"gxt": D, "dht": D, "dab": 2 * Ip, "dqkv": (H + 2 * KV) * (D // H)}.get(f, D)
            out = np.empty((2 * n, cols), np.uint16)
            check(lib().rsys_debug_get(self._h, key.encode(), out.ctypes.data, out.nbytes))
            return (out.astype(np.uint32) << 16).view(np.float32)
        elif key.startswith("f8."):
            L = self.config["num_layers"]
            out = np.empty({"f8.aamax": (L, 64, 32), "f8.wamax": (L, 8), "f8.desc": (L, 8, 32)}[key], np.float32)   # (aamax: [layer][shard][slot], take the max over the shards)
        elif key == "table.fused":
            out = np.empty((V + 1, D), np.float32)
        else:
            is_f32 = key in ("masked.rating", "masked.progress") or key.endswith(".label") or key.endswith(".weight")
            out = np.empty(n, np.float32 if is_f32 else np.int32)
        check(lib().rsys_debug_get(self._h, key.encode(), out.ctypes.data, out.nbytes))
        return out

    def param_checksum(self):
        """[fp64 sum, fp64 sum of squares, low / high 32 bits of a position-weighted integer sum of the bit patterns] of this rank's flat
        parameter buffer (a row-sharded model: without its own table rows), computed on the device in a fixed order: what the ranks
        compare instead of DDP's parameter broadcast (transformer.py:678-682; dist.assert_replicas_equal)"""
        out = (C.c_double * 4)()
        check(lib().rsys_param_checksum(self._h, C.byref(out)))
        return [float(x) for x in out]

    def head_rows(self):
        """positive-weight positions per task in the last forward (the head GEMMs stop there)."""
        out = (C.c_int32 * 4)()
        check(lib().rsys_head_rows_get(self._h, C.byref(out)))
        return [int(x) for x in out]

    # ---- instrumentation
    def timing(self, enable, serialize=False):
        """HIP-event timing of every kernel call site.  serialize=True additionally runs the side-stream GEMMs in
        line, so that each kernel is measured without a concurrent neighbour (bench.py --detail)."""
        check(lib().rsys_op_timing(self._h, (2 if serialize else 1) if enable else 0))

    def timing_pause(self):
        """stop recording call-site events without waiting for the device; timing_report() later returns what was recorded"""
        check(lib().rsys_op_timing(self._h, 3))

    def timing_filter(self, substr):
        """time only the call sites whose name contains `substr` ("" = all); call after timing(True); cleared by timing(False)"""
        check(lib().rsys_op_timing_filter(self._h, (substr or "").encode()))

    def step_mark(self):
        """record a step boundary on the model's stream (no host sync)"""
        check(lib().rsys_step_mark(self._h))

    def step_times_ms(self, cap=65536):
        """milliseconds between consecutive step_mark() calls since the last read (synchronises)"""
        out = np.empty(cap, np.float32); n = C.c_int32()
        check(lib().rsys_step_marks_get(self._h, out.ctypes.data, cap, C.byref(n)))
        return out[: n.value].astype(np.float64)

    def timing_report(self):
        buf = C.create_string_buffer(1 << 16)
        check(lib().rsys_timing_get(self._h, buf, 1 << 16))
        rep = {}
        for line in buf.value.decode().splitlines():
            name, ms, cnt, fl = line.split()
            rep[name] = {"ms": float(ms), "count": int(cnt), "flops": float(fl)}
        return rep


def gather_rows(host_group, rows):
    """the ranks' row blocks of a row-sharded table, concatenated in rank order (over the TCP control plane)"""
    rows = np.ascontiguousarray(rows, np.float32)
    parts = host_group.all_gather_bytes(rows.tobytes())
    return np.concatenate([np.frombuffer(b, np.float32).reshape(-1, rows.shape[1]) for b in parts], axis=0)


def synchronize():
    check(lib().rsys_device_synchronize())
