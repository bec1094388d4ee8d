"""Fixture for the shard writer's event pipeline: the reference's own `tokenize` / `project`
(notebooks/Finetune/embed.py:39-71, the Python mirror of notebooks/Training/history_tools.jl:37-75 `tokenize!` / `project!`)
run on 1 200 synthetic users whose events carry the previous state of the item exactly as the reference's importer leaves
it (`annotate_with_last_state!`, notebooks/Training/import_list.jl:608-635: `nothing` on an item's first event).

TEST INFRASTRUCTURE: run in the build container only (it reads /root/reference); writes tests/golden/tokenize_1k.npz =
the users' raw events and the tokens the reference keeps.  Julia is absent from the image, so the label rules of
`get_data` (transformer.jl:120-139) cannot be executed and stay unpinned.

    python oracle/gen_tokenize_fixture.py
"""
import ast
import os

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
NONE = -99                      # stands for `nothing` / None in the stored arrays
FIELDS = ("medium", "matchedid", "history_max_ts", "status", "rating", "progress", "history_status", "history_rating")


def reference_fns():
    with open("/root/reference/notebooks/Finetune/embed.py") as f:
        tree = ast.parse(f.read())
    ns = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("tokenize", "project"):
            exec(compile(ast.Module([node], []), "embed.py", "exec"), ns)
    return ns["tokenize"], ns["project"]


def make_user(rng, n_events):
    """events in time order; consecutive events often hit the same item (tokenize merges them), some events repeat the
    item's previous state (project drops them), some items come back later"""
    items, ts, snap = [], 9.5e8 + float(rng.integers(0, 10**8)), {}
    mid = None
    seen = []
    for _ in range(n_events):
        r = rng.random()
        if mid is None or r > 0.45:
            if seen and rng.random() < 0.3:
                mid = seen[int(rng.integers(0, len(seen)))]
            else:
                mid = (int(rng.integers(0, 2)), int(rng.integers(1, 300)))
                seen.append(mid)
        ts += float(rng.integers(1, 200000))
        st = int(rng.integers(0, 9))
        rt = float(rng.integers(1, 11)) if rng.random() > 0.45 else 0.0
        if rng.random() < 0.3 and mid in snap:
            st, rt = snap[mid]
        hs, hr = snap.get(mid, (None, None))
        items.append({"medium": mid[0], "matchedid": mid[1], "history_max_ts": ts, "status": st, "rating": rt,
                      "progress": float(rng.integers(0, 65)) / 64.0, "history_status": hs, "history_rating": hr})
        snap[mid] = (st, rt)
    return items


def pack(users):
    off = np.cumsum([0] + [len(u) for u in users]).astype(np.int64)
    cols = {}
    for k in FIELDS:
        vals = [NONE if e[k] is None else e[k] for u in users for e in u]
        cols[k] = np.array(vals, np.float64 if k in ("history_max_ts", "rating", "progress", "history_rating") else np.int32)
    return off, cols


def main():
    tokenize, project = reference_fns()
    rng = np.random.default_rng(20261004)
    users = [make_user(rng, int(rng.integers(1, 80))) for _ in range(1200)]
    users[0] = []                                              # a user with no events
    kept = [project(tokenize([dict(e) for e in u])) for u in users]
    out = {}
    off, cols = pack(users)
    out["in/offsets"] = off
    for k, v in cols.items():
        out["in/" + k] = v
    off, cols = pack(kept)
    out["out/offsets"] = off
    for k, v in cols.items():
        out["out/" + k] = v
    np.savez_compressed(os.path.join(OUT, "tokenize_1k.npz"), **out)
    print("users", len(users), "events", int(out["in/offsets"][-1]), "tokens kept", int(out["out/offsets"][-1]))


if __name__ == "__main__":
    main()
