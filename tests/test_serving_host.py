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


def test_registry_scores():
    """Finetune/embed.jl:86-96 on a small registry."""
    from recommendersystem_amd import serve
    rng = np.random.default_rng(4)
    table = rng.standard_normal((7, 5)).astype(np.float32)
    u = rng.standard_normal(5).astype(np.float32)
    reg = {"1.watch.weight": table, "1.rating_mean": np.float64(7.5)}
    p = serve.compute_retrieval(reg, 1, {"1.retrieval": u})
    z = table.astype(np.float64) @ u
    ref = np.exp(z) / np.exp(z).sum()
    assert p.shape == (7,) and np.allclose(p, ref, rtol=1e-6) and abs(p.sum() - 1) < 1e-6
    assert np.allclose(serve.compute_retrieval(reg, 1, {"1.retrieval": u}, idxs=[2, 0]), ref[[2, 0]], rtol=1e-6)
    reg["1.retrieval.coefs"] = np.array([0.5])
    assert np.allclose(serve.compute_retrieval(reg, 1, {"1.retrieval": u}), 0.5 * ref, rtol=1e-6)
    r = np.array([0.3, -1.0], np.float32)
    assert np.allclose(serve.compute_ranking(reg, 1, {"1.ranking": r}), r)
    reg["1.rating.coefs"] = np.array([1.0, 0.8])
    assert np.allclose(serve.compute_ranking(reg, 1, {"1.ranking": r}), 7.5 + 0.8 * r)
