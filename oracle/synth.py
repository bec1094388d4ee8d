"""Deterministic synthetic configs, parameters and batch records (numpy only).

TEST / BENCH INFRASTRUCTURE.  Shared by the golden-fixture generator
(oracle/gen_golden.py), the numpy oracle tests and bench.py's input generator.
Nothing here is a compute path of the product.

The batch record follows the reference's shard writer
(notebooks/Training/transformer.jl:79-142 `get_data`, :144-163 `concat`):
27 parallel per-interaction arrays; `matchedid` is global (anime offset by the
manga vocab, transformer.jl:114), `.position` is per-medium (:127,:133,:139),
`token_mask_ids`=1 iff the event carries a new rating (:129-130); zero padding
at the tail of a file is the `userid=0` pad user (:146-148).
The corpus statistics follow SURVEY.md section 8(d).
"""
import numpy as np

MEDIUMS = (0, 1)
METRICS3 = ("watch", "rating", "status")
PLANNED_STATUS = 5  # transformer.jl:18

INT_KEYS = ("userid", "token_mask_ids", "gender", "source", "matchedid", "status")
F32_KEYS = ("rating", "progress")


def batch_keys():
    keys = ["userid", "token_mask_ids", "time", "gender", "source", "matchedid",
            "status", "rating", "progress"]
    for m in MEDIUMS:
        for metric in METRICS3:
            keys += [f"{m}.{metric}.label", f"{m}.{metric}.weight", f"{m}.{metric}.position"]
    return keys


def key_dtype(k):
    if k == "time":
        return np.float64
    if k in INT_KEYS or k.endswith(".position"):
        return np.int32
    return np.float32


def make_config(name="tiny", **over):
    """Config dicts with the reference's keys (transformer.py:535-560)."""
    base = {
        "vocab_sizes": {"status": 9, "gender": 4, "source": 4},
        "min_ts": 946684800.0,          # 2000-01-01 UTC
        "max_ts": 1790000000.0,
        "rating_mean": 7.6287384,
        "rating_std": 1.778219,
        "forward": "train",
        "finetune": False,
        "learning_rate": 1e-4,
        "mask_rate": 0.1,
    }
    shapes = {
        # name: L, H, KV, D, I, S, V0, V1, M, K
        "tiny":  (2, 2, 1, 32, 88, 16, 30, 50, 12, 4),       # hd=16
        "hd64":  (2, 2, 1, 128, 352, 64, 120, 200, 20, 12),  # hd=64, GPU-kernel shaped
        "cfg1":  (2, 4, 2, 64, 176, 32, 400, 600, 6148, 8),
        "cfg2":  (8, 4, 2, 256, 704, 256, 60000, 40000, 6148, 32),
        "cfg3":  (8, 8, 4, 512, 1408, 512, 120000, 80000, 6148, 64),
        "cfg4":  (8, 16, 8, 1024, 2816, 512, 120000, 80000, 6148, 64),
    }[name]
    L, H, KV, D, I, S, V0, V1, M, K = shapes
    cfg = dict(base)
    cfg.update({
        "num_layers": L, "num_heads": H, "num_kv_heads": KV, "embed_dim": D,
        "intermediate_dim": I, "max_sequence_length": S,
        "metadata_emb_size": M, "mask_topk": K,
    })
    cfg["vocab_sizes"] = dict(base["vocab_sizes"])
    cfg["vocab_sizes"]["0_matchedid"] = V0
    cfg["vocab_sizes"]["1_matchedid"] = V1
    cfg.update(over)
    return cfg


def param_shapes(cfg):
    """Ordered {state_dict key: shape} of the reference model
    (transformer.model.py:27-49, 97-118, 216-234, 289-295, 312-333, 346-359).
    `watch_head.item_embedding.*` aliases are not listed (shared storage)."""
    D = cfg["embed_dim"]; I = cfg["intermediate_dim"]
    H = cfg["num_heads"]; KV = cfg["num_kv_heads"]; hd = D // H
    V = cfg["vocab_sizes"]["0_matchedid"] + cfg["vocab_sizes"]["1_matchedid"]
    M = cfg["metadata_emb_size"]
    vs = cfg["vocab_sizes"]
    sh = {}
    sh["action_embedding.periodic_time_cos"] = (2,)
    sh["action_embedding.periodic_time_sin"] = (2,)
    sh["action_embedding.status_embedding.embedding.weight"] = (vs["status"] + 1, 16)
    sh["action_embedding.gender_embedding.embedding.weight"] = (vs["gender"] + 1, 4)
    sh["action_embedding.source_embedding.embedding.weight"] = (vs["source"] + 1, 4)
    sh["action_embedding.linear.weight"] = (D, 32)
    sh["action_embedding.linear.bias"] = (D,)
    sh["item_embedding.matchedid_embedding.embedding.weight"] = (V + 1, D)
    sh["item_embedding.metadata_embedding.embedding.weight"] = (V + 1, M)
    sh["item_embedding.projection_layer.weight"] = (D, M)
    sh["item_embedding.projection_layer.bias"] = (D,)
    for l in range(cfg["num_layers"]):
        p = f"transformers.layers.{l}."
        sh[p + "attn.q_proj.weight"] = (H * hd, D)
        sh[p + "attn.k_proj.weight"] = (KV * hd, D)
        sh[p + "attn.v_proj.weight"] = (KV * hd, D)
        sh[p + "attn.output_proj.weight"] = (D, H * hd)
        if cfg.get("finetune"):
            sh[p + "attn.q_proj_lora_A.weight"] = (8, D)
            sh[p + "attn.q_proj_lora_B.weight"] = (H * hd, 8)
            sh[p + "attn.v_proj_lora_A.weight"] = (8, D)
            sh[p + "attn.v_proj_lora_B.weight"] = (KV * hd, 8)
        sh[p + "mlp.w1.weight"] = (I, D)
        sh[p + "mlp.w2.weight"] = (D, I)
        sh[p + "mlp.w3.weight"] = (I, D)
        sh[p + "sa_norm.scale"] = (D,)
        sh[p + "mlp_norm.scale"] = (D,)
    sh["transformers.norm.scale"] = (D,)
    sh["rating_head.0.weight"] = (D, D)
    sh["rating_head.0.bias"] = (D,)
    sh["rating_head.2.weight"] = (1, D)
    sh["rating_head.2.bias"] = (1,)
    return sh


FROZEN = ("item_embedding.metadata_embedding.embedding.weight",)


def trainable_names(cfg):
    names = [n for n in param_shapes(cfg) if n not in FROZEN]
    if cfg.get("finetune"):
        names = [n for n in names if "lora_" in n]  # transformer.model.py:361-371
    return names


def make_params(cfg, seed, style="test"):
    """style="init": the reference init (transformer.model.py:5-12: N(0,0.006),
    zero biases, last row of every embedding zeroed, norm scales 1, phases 0).
    style="test": every tensor random and O(1)-conditioned so that every
    gradient path is exercised (biases, norm scales, phases non-trivial)."""
    rng = np.random.default_rng(seed)
    P = {}
    D = cfg["embed_dim"]
    for name, shape in param_shapes(cfg).items():
        if style == "init":
            if name.endswith(".scale"):
                w = np.ones(shape, np.float32)
            elif name.endswith(".bias") or "periodic" in name:
                w = np.zeros(shape, np.float32)
            else:
                w = (rng.standard_normal(shape) * 0.006).astype(np.float32)
                if "embedding.weight" in name:
                    w[-1] = 0
                if "lora_B" in name:
                    w[:] = 0
        else:
            if name.endswith(".scale"):
                w = (1.0 + 0.2 * rng.standard_normal(shape)).astype(np.float32)
            elif "periodic" in name:
                w = (0.5 * rng.standard_normal(shape)).astype(np.float32)
            elif name.endswith(".bias"):
                w = (0.05 * rng.standard_normal(shape)).astype(np.float32)
            elif "metadata_embedding" in name:
                w = (rng.standard_normal(shape) / np.sqrt(shape[1])).astype(np.float32)
            elif "embedding.weight" in name:
                w = (0.5 * rng.standard_normal(shape)).astype(np.float32)
            else:
                fan_in = shape[-1]
                w = (rng.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
        if "metadata_embedding" in name:
            w[-1] = 0  # transformer.model.py:386 (mask row of the frozen table is zero)
        P[name] = w
    return P


def make_metadata(cfg, seed):
    """Frozen metadata table (V, M) f32, N(0,1)/sqrt(M)."""
    V = cfg["vocab_sizes"]["0_matchedid"] + cfg["vocab_sizes"]["1_matchedid"]
    M = cfg["metadata_emb_size"]
    rng = np.random.default_rng(seed)
    out = np.empty((V, M), np.float32)
    step = 8192
    for i in range(0, V, step):
        n = min(step, V - i)
        out[i:i + n] = rng.standard_normal((n, M), dtype=np.float32) / np.float32(np.sqrt(M))
    return out


def _zipf_ids(rng, n, vmax):
    """Zipf(s=1) over ids 1..vmax-1 by inverse-CDF of the continuous 1/x law."""
    u = rng.random(n)
    ids = np.floor(np.exp(u * np.log(vmax - 1.0))).astype(np.int64)
    return np.clip(ids, 1, vmax - 1)


def make_stream(cfg, n_interactions, seed, mu=4.6, sigma=1.0, min_len=5, max_len=4096):
    """Flat packed stream of user histories (A1 record), zero-padded to
    n_interactions. Targets follow transformer.jl:120-139 with
    history_status/history_rating = previous state of the same item for that
    user (import_list.jl:624-635)."""
    rng = np.random.default_rng(seed)
    V0 = cfg["vocab_sizes"]["0_matchedid"]; V1 = cfg["vocab_sizes"]["1_matchedid"]
    d = {k: np.zeros(n_interactions, key_dtype(k)) for k in batch_keys()}
    pos = 0
    uid = 1
    min_ts, max_ts = cfg["min_ts"], cfg["max_ts"]
    while pos < n_interactions:
        ell = int(np.clip(np.round(rng.lognormal(mu, sigma)), min_len, max_len))
        ell = min(ell, n_interactions - pos)
        sl = slice(pos, pos + ell)
        medium = (rng.random(ell) < 0.7).astype(np.int64)
        mid = np.where(medium == 1, _zipf_ids(rng, ell, V1), _zipf_ids(rng, ell, V0))
        # tokenize!: collapse consecutive events on the same item (history_tools.jl:51-75)
        same = np.zeros(ell, bool)
        same[1:] = (medium[1:] == medium[:-1]) & (mid[1:] == mid[:-1])
        mid = np.where(same, np.clip(mid + 1, 1, np.where(medium == 1, V1, V0) - 1), mid)
        t0 = rng.uniform(min_ts, max_ts - 1.0)
        gaps = rng.exponential((max_ts - t0) / (ell + 1.0), ell)
        times = np.minimum(np.floor(t0 + np.cumsum(gaps)) + rng.random(ell), max_ts)
        status = rng.integers(0, 9, ell)
        rating = np.where(rng.random(ell) < 0.45, 0.0,
                          np.clip(np.round(rng.normal(7.63, 1.78, ell)), 1, 10))
        progress = rng.random(ell)
        d["userid"][sl] = uid
        d["time"][sl] = times
        d["gender"][sl] = rng.choice(4, p=[0.6, 0.25, 0.1, 0.05])
        d["source"][sl] = rng.integers(0, 4)
        d["matchedid"][sl] = mid + np.where(medium == 1, V0, 0)
        d["status"][sl] = status
        d["rating"][sl] = rating
        d["progress"][sl] = progress
        snap = {}
        for i in range(ell):
            m = int(medium[i]); key = (m, int(mid[i]))
            hs, hr = snap.get(key, (None, None))
            st = int(status[i]); rt = float(rating[i])
            inferred = st == 0 and hs is None
            new_watch = st > PLANNED_STATUS and (hs is None or 0 < hs <= PLANNED_STATUS)
            j = pos + i
            if inferred or new_watch:
                d[f"{m}.watch.label"][j] = 1
                d[f"{m}.watch.weight"][j] = 1
                d[f"{m}.watch.position"][j] = mid[i]
            if rt > 0 and rt != hr:
                d["token_mask_ids"][j] = 1
                d[f"{m}.rating.label"][j] = rt
                d[f"{m}.rating.weight"][j] = 1
                d[f"{m}.rating.position"][j] = mid[i]
            if st > 0 and st != hs:
                d[f"{m}.status.label"][j] = st
                d[f"{m}.status.weight"][j] = 1
                d[f"{m}.status.position"][j] = mid[i]
            snap[key] = (st, rt)
        pos += ell
        uid += 1
    return d


def make_batch(cfg, rows, seed, **kw):
    """One (rows*S,) flat batch, as PretrainDataset yields (transformer.py:91-98)."""
    S = cfg["max_sequence_length"]
    kw.setdefault("mu", np.log(max(6.0, S / 4.0)))
    kw.setdefault("sigma", 0.8)
    return make_stream(cfg, rows * S, seed, **kw)


def make_masks(cfg, rows, seed):
    """Pretraining masks from a uniform draw (transformer.model.py:437-440)."""
    S = cfg["max_sequence_length"]
    rng = np.random.default_rng(seed)
    u = rng.random((rows, S)).astype(np.float32)
    r = np.float32(cfg["mask_rate"])
    watch = u < r
    rating = (u >= r) & (u < 2 * r)
    return watch, rating
