"""Host side of the inference forward: the request -> batch -> model -> response steps of the
reference's embedding server (notebooks/Finetune/embed.py:27-161), on top of `RecommenderModel.inference_forward`
(rsys_infer: fused item table, per-row `rope_input_pos`, per-candidate `token_mask_ids`).

A user's history is a list of events {"medium", "matchedid", "history_max_ts", "status", "rating", "progress",
"history_status", "history_rating"}; `tokenize` merges consecutive events on the same item (the item keeps the first
event's identity and the last event's state, embed.py:39-60), `project` drops events that did not change status or rating
(embed.py:63-71).  Retrieval appends one query token (item -1) and returns the trunk output at that item token; ranking
appends one token per candidate item, each with its own `token_mask_ids` value (candidates see the history but not each
other), and returns the rating head at the candidates' action tokens.
"""
import numpy as np


_STATE_KEYS = ("status", "rating", "progress")
_INT_COLS = ("userid", "rope_input_pos", "token_mask_ids", "gender", "source", "matchedid", "status")


def make_item(ts, medium=0, itemid=-1):
    """A query / candidate token: item `itemid` of `medium` at time `ts` with the masked action state (embed.py:27-36)."""
    return dict(medium=medium, history_max_ts=ts, matchedid=itemid, status=-1, rating=0, progress=0)


def tokenize(user_items):
    """Runs of consecutive events on the same (medium, item) collapse into one token that keeps the first event's fields
    and takes status / rating / progress from the last one (embed.py:39-60)."""
    from itertools import groupby
    tokens = []
    for _, run in groupby(user_items, key=lambda e: (e["medium"], e["matchedid"])):
        run = list(run)
        tok = dict(run[0])
        tok.update({k: run[-1][k] for k in _STATE_KEYS})
        tokens.append(tok)
    return tokens


def project(user_items):
    """Keeps the tokens whose status or rating differs from the item's previous state (embed.py:63-71)."""
    changed = lambda e: e["history_status"] != e["status"] or e["history_rating"] != e["rating"]
    return [e for e in user_items if changed(e)]


def _history(user, max_user_len):
    """projected tokens of a user, newest `max_user_len - 1` kept (one slot is reserved for the query, embed.py:103-106)."""
    hist = project(tokenize(user["items"]))
    return hist[-(max_user_len - 1):] if len(hist) > max_user_len - 1 else hist


def build_batch(users, task, medium, num_items_0, max_user_len=1024, max_ranking_items=1024):
    """The ten (len(users), max_seq_len) arrays of one inference request (embed.py:74-138): history tokens first, then the
    query token (retrieval) or one token per candidate (ranking).  History tokens get positions 0..n-1 and mask id 0; every
    appended token sits at position n, and ranking candidates carry their own index as `token_mask_ids`, which hides them
    from each other (model.py:479-487)."""
    if task not in ("retrieval", "ranking"):
        raise AssertionError(task)
    width = max_user_len + (max_ranking_items if task == "ranking" else 0)
    n = len(users)
    d = {k: np.zeros((n, width), np.int32) for k in _INT_COLS}
    d["time"] = np.zeros((n, width), np.float64)
    d["rating"] = np.zeros((n, width), np.float32)
    d["progress"] = np.zeros((n, width), np.float32)
    for row, u in enumerate(users):
        hist = _history(u, max_user_len)
        if task == "ranking":
            tail = [make_item(u["timestamp"], medium, cand) for cand in u["ranking_items"]]
        else:
            tail = [make_item(u["timestamp"])]
        seq = hist + tail
        L, nh = len(seq), len(hist)
        who = u["user"]
        d["userid"][row, :L] = row + 1
        d["gender"][row, :L] = 0 if who["gender"] is None else who["gender"] + 1
        d["source"][row, :L] = who["source"]
        d["time"][row, :L] = [e["history_max_ts"] for e in seq]
        d["rope_input_pos"][row, :L] = np.minimum(np.arange(L), nh)
        if task == "ranking":
            d["token_mask_ids"][row, nh:L] = np.arange(nh, L)
        d["matchedid"][row, :L] = [e["matchedid"] + (num_items_0 if e["medium"] == 1 else 0) for e in seq]
        d["status"][row, :L] = [e["status"] for e in seq]
        d["rating"][row, :L] = [e["rating"] for e in seq]
        d["progress"][row, :L] = [e["progress"] for e in seq]
    return d


def extract(embs, users, task, medium, max_user_len=1024):
    """Per user, the rows of the model output that answer the request (embed.py:147-161): token 2n is the query's item token
    (retrieval: its trunk output), tokens 2(n+j)+1 are the candidates' action tokens (ranking: their rating-head value)."""
    key = f"{medium}.{task}"
    embs = np.asarray(embs)
    out = []
    for row, u in enumerate(users):
        n = len(_history(u, max_user_len))
        if task == "retrieval":
            out.append({key: embs[row, 2 * n, :].tolist()})
        else:
            rows = 2 * (n + np.arange(len(u["ranking_items"]))) + 1
            out.append({key: embs[row, rows, 0].tolist()})
    return out


def predict(model, users, task, medium, max_user_len=None, max_ranking_items=None):
    """embed.py:74-161 on the HIP model (`model.config["forward"]` semantics = inference): sequence length of the request =
    the model's `max_sequence_length` (retrieval: all of it is history + query; ranking: split between history and candidates)."""
    S = model.config["max_sequence_length"]
    if task == "retrieval":
        max_user_len = S if max_user_len is None else max_user_len
        max_ranking_items = 0
        assert max_user_len == S
    else:
        max_user_len = S // 2 if max_user_len is None else max_user_len
        max_ranking_items = S - max_user_len if max_ranking_items is None else max_ranking_items
        assert max_user_len + max_ranking_items == S
    d = build_batch(users, task, medium, model.config["vocab_sizes"]["0_matchedid"], max_user_len, max_ranking_items)
    if not hasattr(model, "inference_select"):           # (a model that only has the reference's call: the full tensor, then extract)
        return extract(model.inference_forward(d, task), users, task, medium, max_user_len)
    # only the tokens `extract` would read leave the device (one row per user for retrieval, the candidates' action tokens for
    # ranking) instead of the (rows, 2S, D) tensor
    index, counts = [], []
    for row, u in enumerate(users):
        n = len(_history(u, max_user_len))
        toks = [2 * n] if task == "retrieval" else list(2 * (n + np.arange(len(u["ranking_items"]))) + 1)
        index += [row * 2 * S + int(t) for t in toks]; counts.append(len(toks))
    key = f"{medium}.{task}"
    if not index:
        return [{key: []} for _ in users]
    vals = model.inference_select(d, task, index)
    out, at = [], 0
    for c in counts:
        out.append({key: (vals[at].tolist() if task == "retrieval" else vals[at:at + c].tolist())}); at += c
    return out


def register_transformer(model, path):
    """Finetune/register.py:14-36: the serving registry `model.registry.h5` -- the item table split by medium (the
    retrieval scores are softmax(table . user embedding), Finetune/embed.jl:86-90) and the rating offset of each medium."""
    from . import h5
    n0 = model.config["vocab_sizes"]["0_matchedid"]
    embs = model.item_embeddings()
    mean = np.float64(model.config["rating_mean"])
    h5.write_h5(path, {"0.watch.weight": embs[:n0], "1.watch.weight": embs[n0:], "0.rating_mean": mean, "1.rating_mean": mean}, blosc=None)


def compute_retrieval(registry, medium, user, idxs=None):
    """Finetune/embed.jl:86-90: p = softmax(table_m . u) over the medium's items (optionally at `idxs`), times the
    registry's retrieval coefficient when it has one (`{m}.retrieval.coefs`, fitted by Finetune/regress.jl)."""
    logits = np.asarray(registry[f"{medium}.watch.weight"], np.float64) @ np.asarray(user[f"{medium}.retrieval"], np.float64)
    p = np.exp(logits - logits.max())
    p /= p.sum()
    if idxs is not None:
        p = p[np.asarray(idxs)]
    coefs = registry.get(f"{medium}.retrieval.coefs")
    return (p * np.asarray(coefs).reshape(-1)[0] if coefs is not None else p).astype(np.float32)


def compute_ranking(registry, medium, user):
    """Finetune/embed.jl:92-96: blend of the medium's mean rating and the model's rating predictions with the registry's
    coefficients (`{m}.rating.coefs` = [baseline, model]); without coefficients the model's prediction alone."""
    r_masked = np.asarray(user[f"{medium}.ranking"], np.float64)
    coefs = registry.get(f"{medium}.rating.coefs")
    if coefs is None:
        return r_masked.astype(np.float32)
    c = np.asarray(coefs, np.float64).reshape(-1)
    return (c[0] * float(np.asarray(registry[f"{medium}.rating_mean"])) + c[1] * r_masked).astype(np.float32)
