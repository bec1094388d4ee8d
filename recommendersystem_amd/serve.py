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


def make_item(ts, medium=0, itemid=-1):   # embed.py:27-36
    return {"medium": medium, "history_max_ts": ts, "matchedid": itemid, "status": -1, "rating": 0, "progress": 0}


def tokenize(user_items):   # embed.py:39-60
    def span_to_token(x):
        token = x[0].copy()
        for k in ["status", "rating", "progress"]:
            token[k] = x[-1][k]
        return token

    items, last_mid, span = [], None, []
    for x in user_items:
        mid = (x["medium"], x["matchedid"])
        if mid == last_mid:
            span.append(x)
        else:
            if span:
                items.append(span_to_token(span))
            span = [x]
            last_mid = mid
    if span:
        items.append(span_to_token(span))
    return items


def project(user_items):   # embed.py:63-71
    return [x for x in user_items
            if not ((x["history_status"] == x["status"]) and (x["history_rating"] == x["rating"]))]


def build_batch(users, task, medium, num_items_0, max_user_len=1024, max_ranking_items=1024):
    """embed.py:74-138: the ten (len(users), max_seq_len) arrays of one inference request."""
    assert task in ("retrieval", "ranking")
    max_seq_len = max_user_len if task == "retrieval" else max_user_len + max_ranking_items
    n = len(users)
    d = {"userid": np.zeros((n, max_seq_len), np.int32), "time": np.zeros((n, max_seq_len), np.float64),
         "rope_input_pos": np.zeros((n, max_seq_len), np.int32), "token_mask_ids": np.zeros((n, max_seq_len), np.int32),
         "gender": np.zeros((n, max_seq_len), np.int32), "source": np.zeros((n, max_seq_len), np.int32),
         "matchedid": np.zeros((n, max_seq_len), np.int32), "status": np.zeros((n, max_seq_len), np.int32),
         "rating": np.zeros((n, max_seq_len), np.float32), "progress": np.zeros((n, max_seq_len), np.float32)}
    for u in range(n):
        user = users[u]["user"]
        items = project(tokenize(users[u]["items"]))
        extra_tokens = 1
        if len(items) > max_user_len - extra_tokens:
            items = items[-(max_user_len - extra_tokens):]
        if task == "ranking":
            test_items = [make_item(users[u]["timestamp"], medium, x) for x in users[u]["ranking_items"]]
        else:
            test_items = [make_item(users[u]["timestamp"])]
        for i, x in enumerate(items + test_items):
            d["userid"][u, i] = u + 1
            d["time"][u, i] = x["history_max_ts"]
            d["gender"][u, i] = 0 if user["gender"] is None else user["gender"] + 1
            d["source"][u, i] = user["source"]
            d["rope_input_pos"][u, i] = i if i < len(items) else len(items)
            d["token_mask_ids"][u, i] = i if task == "ranking" and i >= len(items) else 0
            d["matchedid"][u, i] = x["matchedid"] + (num_items_0 if x["medium"] == 1 else 0)
            d["status"][u, i] = x["status"]
            d["rating"][u, i] = x["rating"]
            d["progress"][u, i] = x["progress"]
    return d


def extract(embs, users, task, medium, max_user_len=1024):
    """embed.py:147-161: the query item token (retrieval) / the candidates' action tokens (ranking) of every row."""
    ret = []
    for i, u in enumerate(users):
        N = min(len(project(tokenize(u["items"]))), max_user_len - 1)
        if task == "retrieval":
            ret.append({f"{medium}.{task}": np.asarray(embs[i, 2 * N, :]).tolist()})
        else:
            idxs = [2 * (N + j) + 1 for j in range(len(u["ranking_items"]))]
            ret.append({f"{medium}.{task}": np.asarray(embs[i, idxs, 0]).tolist()})
    return ret


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
    embs = model.inference_forward(d, task)
    return extract(embs, users, task, medium, max_user_len)
