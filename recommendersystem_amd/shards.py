"""Shard writer: per-user msgpack histories -> flat per-token HDF5 shards.

Restates the data-preparation step the reference runs in Julia before training (notebooks/Training/transformer.jl,
history_tools.jl); Julia is not in this image, so this module is checked against hand-derived cases and the format
round trip only (tests/test_shards.py) -- "parity unpinned" for the label rules, pinned for tokenize / project, which
the reference also has in Python (Finetune/embed.py:39-71, see serve.py).

    get_data            transformer.jl:79-146    one user -> the 27 per-token arrays
    concat              transformer.jl:148-168   users of one part, zero-padded to a whole number of batches
    save_data           transformer.jl:202-240   users -> {datadir}/{split}/{shard}/{p}.h5, num_tokens.txt
    pad_splits          transformer.jl:181-200   pad.h5 so that every shard holds the same number of tokens
    save_media_embeddings  transformer.jl:56-77  media_embeddings.h5 ("metadata": text | image | 4 date features)
    get_finetune_data / save_finetune_data   notebooks/Finetune/transformer.jl:52-166   one user per row of 1024 tokens:
                        history, then the held-out test event, which alone carries targets (weights normalised per task)

The random choices (mini subsample, duplicate users to fill the last round of shards, shuffles) use a numpy generator;
they are statistically, not bitwise, those of the Julia run.
"""
import datetime
import glob
import os

import numpy as np

from . import h5
from .serve import project, tokenize

MEDIUMS = (0, 1)
METRICS = ("watch", "rating", "status")
PLANNED_STATUS = 5                      # transformer.jl:19
BATCH_SIZE = 128 * 1024                 # transformer.jl:21: local_batch_size * max_sequence_length
NUM_GPUS = 8                            # transformer.jl:22
MIN_TS = datetime.datetime(2000, 1, 1, tzinfo=datetime.timezone.utc).timestamp()   # transformer.jl:23
USERS_PER_PART = 65_536                 # transformer.jl:217

_INT_KEYS = ("userid", "token_mask_ids", "gender", "source", "matchedid", "status")


def max_ts_of(list_tag):
    """transformer.jl:24-26: `list_tag` is the yyyymmdd snapshot tag."""
    return datetime.datetime.strptime(list_tag.strip(), "%Y%m%d").replace(tzinfo=datetime.timezone.utc).timestamp()


def optdate(x, max_ts, min_ts=MIN_TS):
    """transformer.jl:38-54: (present, normalised date); a date that does not parse loses trailing fields until it does."""
    if x is None or x == "":
        return 0, 0.0
    fields = str(x).split("-")
    try:
        ymd = [int(f) for f in fields]
        if not 1 <= len(ymd) <= 3:
            raise ValueError(x)
        ymd += [1] * (3 - len(ymd))
        ts = datetime.datetime(*ymd, tzinfo=datetime.timezone.utc).timestamp()
        y = (ts - min_ts) / (max_ts - min_ts)
        return 1, float(min(max(y, -5.0), 5.0))
    except ValueError:
        if len(fields) > 1:
            return optdate("-".join(fields[:-1]), max_ts, min_ts)
    return 0, 0.0


def media_embedding_matrix(media, num_items, max_ts, text_dim=3072, image_dim=3072):
    """transformer.jl:56-71.  `media[m]` = records of medium m with `matchedid`, `text_embedding.embedding`,
    `image_embedding`, `metadata.dates.{startdate,enddate}`; returns the (V, M) float32 table h5py shows the reference
    (the Julia array is its (M, V) transpose), V = num_items[0] + num_items[1], M = text + image + 4."""
    W = np.zeros((sum(num_items[m] for m in MEDIUMS), text_dim + image_dim + 4), np.float32)
    for m in MEDIUMS:
        for x in media.get(m, []):
            has_sd, sd = optdate(x["metadata"]["dates"]["startdate"], max_ts)
            has_ed, ed = optdate(x["metadata"]["dates"]["enddate"], max_ts)
            row = x["matchedid"] + (num_items[0] if m == 1 else 0)
            W[row] = np.concatenate([np.asarray(x["text_embedding"]["embedding"], np.float32),
                                     np.asarray(x["image_embedding"], np.float32), [has_sd, sd, has_ed, ed]])
    return W


def save_media_embeddings(datadir, media, num_items, max_ts, **dims):
    """transformer.jl:72-77."""
    h5.write_h5(os.path.join(datadir, "media_embeddings.h5"), {"metadata": media_embedding_matrix(media, num_items, max_ts, **dims)}, blosc=3)


def _empty_record(n):
    """the 27 arrays of the batch record, all zero, for n tokens"""
    d = {k: np.zeros(n, np.int32) for k in _INT_KEYS}
    d["time"] = np.zeros(n, np.float64)
    d["rating"] = np.zeros(n, np.float32)
    d["progress"] = np.zeros(n, np.float32)
    for m in MEDIUMS:
        for metric in METRICS:
            d[f"{m}.{metric}.label"] = np.zeros(n, np.float32)
            d[f"{m}.{metric}.weight"] = np.zeros(n, np.float32)
            d[f"{m}.{metric}.position"] = np.zeros(n, np.int32)
    return d


def _event_columns(events):
    """The events of one user as columns.  `prev_status` / `prev_rating`: the state of the same item before the event
    (`history_status` / `history_rating` of the importer, import_list.jl:624-635), NaN where the item is new to the user."""
    col = lambda key, dt: np.array([x[key] for x in events], dt) if events else np.zeros(0, dt)
    prev = lambda key, dt: np.array([np.nan if x[key] is None else x[key] for x in events], dt) if events else np.zeros(0, dt)
    # prev_rating in the precision of `rating` (both Float32 in transformer.jl:129): 8.3 must compare equal to a history rating of 8.3
    return {"medium": col("medium", np.int64), "matchedid": col("matchedid", np.int64), "status": col("status", np.int64),
            "rating": col("rating", np.float32), "progress": col("progress", np.float32), "time": col("history_max_ts", np.float64),
            "prev_status": prev("history_status", np.float64), "prev_rating": prev("history_rating", np.float32)}


def _target_masks(c):
    """The three target rules of the writer (transformer.jl:120-139) as boolean columns over the events:
      watch   the event is an implied watch (status 0 on an item the user has no state for), or the item's status rises above
              the planned states from no state / a planned state;
      rating  the event carries a rating that differs from the item's previous one;
      status  the event carries a status that differs from the item's previous one."""
    new_item = np.isnan(c["prev_status"])
    was_planned = (c["prev_status"] > 0) & (c["prev_status"] <= PLANNED_STATUS)      # (NaN compares false)
    watch = ((c["status"] == 0) & new_item) | ((c["status"] > PLANNED_STATUS) & (new_item | was_planned))
    rating = (c["rating"] > 0) & ~(c["rating"] == c["prev_rating"])                  # float32 against float32 (NaN: new item)
    status = (c["status"] > 0) & ~(c["status"].astype(np.float64) == c["prev_status"])
    return {"watch": watch, "rating": rating, "status": status}


def _write_events(d, at, c, user, userid, num_items_0, with_targets):
    """events `c` into tokens at .. at + n of record `d`; targets (label, weight 1, per-medium position) where asked"""
    n = len(c["medium"])
    sl = slice(at, at + n)
    d["userid"][sl] = userid
    d["time"][sl] = c["time"]
    d["gender"][sl] = 0 if user["gender"] is None else user["gender"] + 1      # 0 = unknown (transformer.jl:111)
    d["source"][sl] = user["source"]
    d["matchedid"][sl] = c["matchedid"] + num_items_0 * (c["medium"] == 1)     # global id: anime behind manga (:114)
    d["status"][sl] = c["status"]
    d["rating"][sl] = c["rating"]
    d["progress"][sl] = c["progress"]
    if not with_targets:
        return
    masks = _target_masks(c)
    labels = {"watch": np.ones(n, np.float32), "rating": c["rating"], "status": c["status"].astype(np.float32)}
    d["token_mask_ids"][sl] = masks["rating"]                                  # tokens whose rating may be hidden (model.py:479-487)
    for m in MEDIUMS:
        in_m = c["medium"] == m
        for metric in METRICS:
            hit = masks[metric] & in_m
            d[f"{m}.{metric}.label"][sl] = np.where(hit, labels[metric], 0)
            d[f"{m}.{metric}.weight"][sl] = hit
            d[f"{m}.{metric}.position"][sl] = np.where(hit, c["matchedid"], 0)


def get_data(data, userid, num_items_0):
    """transformer.jl:79-146.  `data` = {"user": {...}, "items": [events]}; returns the 27 arrays of this user's tokens: the
    projected events as columns, every event a candidate target (rules: _target_masks)."""
    c = _event_columns(project(tokenize(data["items"])))
    d = _empty_record(len(c["medium"]))
    _write_events(d, 0, c, data["user"], userid, num_items_0, with_targets=True)
    return d


def concat(ds, batch_size=BATCH_SIZE):
    """transformer.jl:148-168: users back to back, zero tokens up to the next multiple of `batch_size`."""
    n = sum(len(x["userid"]) for x in ds)
    total = n + (-n) % batch_size
    out = {}
    for k, v in ds[0].items():
        out[k] = np.zeros(total, v.dtype)
        out[k][:n] = np.concatenate([x[k] for x in ds])
    return out


def get_num_tokens(splitdir, shard):
    """transformer.jl:170-179."""
    tokens = 0
    for fn in glob.glob(os.path.join(splitdir, str(shard), "*.h5")):
        with h5.File(fn) as f:
            tokens += f.info("userid")[1][0]
    return tokens


def pad_splits(splitdir, num_shards):
    """transformer.jl:181-200: shards short of the longest one get a pad.h5 whose first tokens repeat the start of their
    1.h5 and whose remainder is zero (zero tokens carry zero target weights)."""
    counts = [get_num_tokens(splitdir, s) for s in range(1, num_shards + 1)]
    for s, have in zip(range(1, num_shards + 1), counts):
        num_padding = max(counts) - have
        if num_padding == 0:
            continue
        with h5.File(os.path.join(splitdir, str(s), "1.h5")) as f:
            out = {}
            for k in f:
                v = f[k]
                n = min(num_padding, len(v))
                out[k] = np.zeros(num_padding, v.dtype)
                out[k][:n] = v[:n]
        h5.write_h5(os.path.join(splitdir, str(s), "pad.h5"), out, blosc=3)


def save_data(datadir, datasplit, transdir, num_items_0, mini=False, num_shards=NUM_GPUS, batch_size=BATCH_SIZE,
              users_per_part=USERS_PER_PART, seed=0, load=None):
    """transformer.jl:202-240: `{datadir}/users/{datasplit}/*/*.msgpack` -> `{datadir}/{transdir}/{datasplit}/{shard}/{p}.h5`
    (+ pad.h5, num_tokens.txt).  Users are dealt round-robin to `num_shards` shards after a shuffle; `userid` is the
    user's 1-based index inside its part, which is all the block shuffle of the reader needs (train.py:53-73)."""
    if load is None:
        import msgpack

        def load(fn):
            with open(fn, "rb") as f:
                return msgpack.unpackb(f.read(), raw=False, strict_map_key=False)
    rng = np.random.default_rng(seed)
    users = sorted(glob.glob(os.path.join(datadir, "users", datasplit, "*", "*.msgpack")))
    if mini:
        users = [x for x in users if rng.random() < 0.5]
    assert users, f"no user files under {datadir}/users/{datasplit}"
    while len(users) % num_shards != 0:
        users.append(users[rng.integers(len(users))])
    users = [users[i] for i in rng.permutation(len(users))]
    splitdir = os.path.join(datadir, transdir, datasplit)
    for shard in range(1, num_shards + 1):
        dest = os.path.join(splitdir, str(shard))
        os.makedirs(dest, exist_ok=True)
        files = [x for i, x in enumerate(users, start=1) if i % num_shards + 1 == shard]
        files = [files[i] for i in rng.permutation(len(files))]
        for p, lo in enumerate(range(0, len(files), users_per_part), start=1):
            part = files[lo:lo + users_per_part]
            ds = [get_data(load(fn), i, num_items_0) for i, fn in enumerate(part, start=1)]
            h5.write_h5(os.path.join(dest, f"{p}.h5"), concat(ds, batch_size), blosc=3)
    pad_splits(splitdir, num_shards)
    total = sum(get_num_tokens(splitdir, s) for s in range(1, num_shards + 1))
    with open(os.path.join(splitdir, "num_tokens.txt"), "w") as f:
        f.write(str(total))
    return total


# ---- finetune shards (notebooks/Finetune/transformer.jl) -----------------------------------------------------------

FINETUNE_SEQ_LEN = 1024                 # Finetune/transformer.jl:21
NUM_TEST_ITEMS = 1                      # :55


def get_finetune_data(data, userid, num_items_0, max_seq_len=FINETUNE_SEQ_LEN):
    """Finetune/transformer.jl:52-133.  `data` = {"user", "items": history events, "test_items": [<= 1 held-out event]}.
    One row of `max_seq_len` tokens: the newest max_seq_len - 1 projected history tokens (context: no targets), then the test
    event, which alone is a target (same rules as the pretraining writer); every task's weights are normalised to sum 1."""
    assert len(data["test_items"]) <= NUM_TEST_ITEMS
    history = project(tokenize(data["items"]))
    keep = max_seq_len - NUM_TEST_ITEMS
    history = history[len(history) - keep:] if keep > 0 and len(history) > keep else (history if keep > 0 else [])   # ([-0:] is the whole list)
    d = _empty_record(max_seq_len)
    _write_events(d, 0, _event_columns(history), data["user"], userid, num_items_0, with_targets=False)
    _write_events(d, len(history), _event_columns(data["test_items"]), data["user"], userid, num_items_0, with_targets=True)
    for m in MEDIUMS:
        for metric in METRICS:
            w = d[f"{m}.{metric}.weight"]
            if w.sum() > 0:
                w /= w.sum()
    return d


def save_finetune_data(datadir, datasplit, num_items_0, num_shards=1, users_per_part=USERS_PER_PART, max_seq_len=FINETUNE_SEQ_LEN,
                       seed=0, load=None):
    """Finetune/transformer.jl:135-166: `{datadir}/users/{datasplit}/*/*.msgpack` -> `{datadir}/transformer/{datasplit}/{shard}/{p}.h5`
    with every dataset of shape (users, max_seq_len) as h5py shows it (Julia stacks the users as columns)."""
    if load is None:
        import msgpack

        def load(fn):
            with open(fn, "rb") as f:
                return msgpack.unpackb(f.read(), raw=False, strict_map_key=False)
    rng = np.random.default_rng(seed)
    users = sorted(glob.glob(os.path.join(datadir, "users", datasplit, "*", "*.msgpack")))
    assert users, f"no user files under {datadir}/users/{datasplit}"
    while len(users) % num_shards != 0:
        users.append(users[rng.integers(len(users))])
    users = [users[i] for i in rng.permutation(len(users))]
    n_rows = 0
    for shard in range(1, num_shards + 1):
        dest = os.path.join(datadir, "transformer", datasplit, str(shard))
        os.makedirs(dest, exist_ok=True)
        files = [x for i, x in enumerate(users, start=1) if i % num_shards + 1 == shard]
        files = [files[i] for i in rng.permutation(len(files))]
        for p, lo in enumerate(range(0, len(files), users_per_part), start=1):
            part = files[lo:lo + users_per_part]
            rows = [get_finetune_data(load(fn), i, num_items_0, max_seq_len) for i, fn in enumerate(part, start=1)]
            rows = [rows[i] for i in rng.permutation(len(rows))]
            h5.write_h5(os.path.join(dest, f"{p}.h5"), {k: np.stack([r[k] for r in rows]) for k in rows[0]}, blosc=3)
            n_rows += len(rows)
    return n_rows
