"""CPU: host side of the inference request path (recommendersystem_amd/serve.py) against the reference's own
`predict` (notebooks/Finetune/embed.py:74-161, run by oracle/gen_golden.py with a recording stand-in for the model):
same batch arrays, same output extraction."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_build_batch_and_extract_match_the_reference_server():
    from recommendersystem_amd import serve
    z = np.load(os.path.join(GOLDEN, "serving.npz"))
    users = json.loads(bytes(z["users_json"]).decode())
    n0 = int(z["num_items_0"][0])
    for task, medium in (("retrieval", 1), ("ranking", 0)):
        d = serve.build_batch(users, task, medium, n0, 1024, 1024)
        keys = sorted(k.split("/")[-1] for k in z.files if k.startswith(f"{task}/in/"))
        assert sorted(d.keys()) == keys
        for k in keys:
            ref = z[f"{task}/in/{k}"]
            assert d[k].dtype == ref.dtype and d[k].shape == ref.shape, k
            np.testing.assert_array_equal(d[k], ref, err_msg=f"{task} {k}")
        n, L = d["userid"].shape
        width = 3 if task == "retrieval" else 1
        embs = (np.arange(n * 2 * L * width, dtype=np.float32) * 0.5).reshape(n, 2 * L, width)   # what the stand-in returned
        assert serve.extract(embs, users, task, medium, 1024) == json.loads(bytes(z[f"{task}/ret_json"]).decode())


def test_tokenize_and_project_properties():
    from recommendersystem_amd import serve
    ev = lambda m, i, st, rt, hs, hr: {"medium": m, "matchedid": i, "history_max_ts": 1.0, "status": st, "rating": rt,
                                        "progress": 0.5, "history_status": hs, "history_rating": hr}
    items = [ev(0, 5, 1, 0, -1, 0), ev(0, 5, 2, 7, 1, 0), ev(1, 5, 3, 0, -1, 0), ev(0, 5, 2, 7, 2, 7)]
    tok = serve.tokenize(items)
    assert [(t["medium"], t["matchedid"], t["status"], t["rating"]) for t in tok] == [(0, 5, 2, 7), (1, 5, 3, 0), (0, 5, 2, 7)]
    assert tok[0]["history_status"] == -1                    # identity of the first event of the span, state of the last
    assert len(serve.project(tok)) == 2                      # the last token changed nothing
