"""Workloads: the configurations of SURVEY 8 (cfg-1 .. cfg-4 and two test sizes) and the synthetic interaction corpus
of SURVEY 8(d), as batch records in the reference's shard layout (numpy only; inputs, no model arithmetic).

The batch record follows the reference's shard writer (notebooks/Training/transformer.jl:79-142 `get_data`,
:144-163 `concat`): 27 parallel per-interaction arrays; `matchedid` is global (anime offset by the manga vocab,
transformer.jl:114), `.position` is per-medium (:127,:133,:139), `token_mask_ids` = 1 iff the event carries a new
rating (:129-130); zero padding at the tail of a file is the `userid = 0` pad user (:146-148).  Users: log-normal
history lengths, Zipf item popularity, 70 % anime, strictly increasing times; targets by the writer's rules with
history_status / history_rating = the previous state of the same item for that user.  bench.py, the smoke test and the
test suite draw their inputs here.
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
        "f8t":   (2, 4, 2, 256, 384, 64, 120, 200, 20, 12),  # smallest shape the fp8 trunk takes (K tiles of 128, kv group of 128 columns)
        "cfg1":  (2, 4, 2, 64, 176, 32, 400, 600, 6148, 8),
        "cfg2":  (8, 4, 2, 256, 704, 256, 60000, 40000, 6148, 32),
        "cfg3":  (8, 8, 4, 512, 1408, 512, 120000, 80000, 6148, 64),
        "cfg4":  (8, 16, 8, 1024, 2816, 512, 120000, 80000, 6148, 64),
        # the reference's production shape (transformer.py:535-560; vocabulary as in cfg-3, the real one comes from {manga,anime}.csv)
        "prod":  (8, 32, 16, 2048, 5632, 1024, 120000, 80000, 6148, 128),
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
