"""CPU: the HDF5 + blosc shard adapter (include/rsys_h5.h, recommendersystem_amd/h5.py) and the shard writer
(recommendersystem_amd/shards.py, restating notebooks/Training/transformer.jl:38-240).

Format parity is checked against an independent client where the image has one: h5py 3.3 under /opt/conda's Python 3.9
(the reference's reader is h5py + hdf5plugin, transformer.py:10-11, 86-89) with this repo's filter plugin on
HDF5_PLUGIN_PATH.  The label rules of `get_data` exist only in Julia in the reference and Julia is absent, so those
cases are hand-derived from transformer.jl:117-143."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H5PY_PYTHON = "/opt/conda/bin/python3.9"


@pytest.fixture(scope="module")
def h5():
    from recommendersystem_amd import h5 as mod
    if not os.path.exists(mod.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    if not os.path.exists(mod.LIB_PATH):
        pytest.skip("no libhdf5 in this image: librsys_h5.so not built")
    return mod


def _sample(rng, n=50_000):
    return {"userid": np.repeat(np.arange(n // 50, dtype=np.int32), 50), "time": rng.random(n),
            "rating": rng.integers(0, 11, n).astype(np.float32), "0.watch.position": rng.integers(0, 9000, n).astype(np.int32),
            "matrix": rng.standard_normal((37, 260)).astype(np.float32), "empty": np.zeros(0, np.float32),
            "noise": rng.integers(0, 256, 1 << 20, dtype=np.uint8), "wide": rng.integers(-9, 9, 4000, dtype=np.int64)}


def test_header_symbols_exported(h5):
    hdr = open(os.path.join(ROOT, "include", "rsys_h5.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(rsys_h5_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(h5.EXPORTED)
    for name in declared:
        assert hasattr(h5.lib(), name), name


def test_round_trip_and_errors(h5, tmp_path):
    d = _sample(np.random.default_rng(1))
    fn = str(tmp_path / "a.h5")
    h5.write_h5(fn, d, blosc=3)
    assert os.path.getsize(fn) < sum(v.nbytes for v in d.values())        # blosc did something
    with h5.File(fn) as f:
        assert f.keys() == sorted(d)                                      # HDF5 name order, like `for k in f`
        for k, v in d.items():
            dt, shape, level = f.info(k)
            assert (dt, shape) == (v.dtype, v.shape) and level == (3 if v.size else -1), k
            assert np.array_equal(f[k], v), k
        with pytest.raises(h5.H5Error):
            f["missing"]
        with pytest.raises(h5.H5Error):
            f.write("x", np.zeros(3, np.float32))                          # read-only handle
    h5.write_h5(fn, {"a": np.arange(5, dtype=np.int32)}, blosc=None)       # contiguous, truncates the file
    with h5.File(fn) as f:
        assert f.keys() == ["a"] and f.info("a")[2] == -1
    with pytest.raises(h5.H5Error):
        h5.File(str(tmp_path / "nope.h5"))
    (tmp_path / "junk.h5").write_bytes(b"not an hdf5 file" * 100)
    with pytest.raises(h5.H5Error):
        h5.read_h5(str(tmp_path / "junk.h5"))


def test_corrupt_chunk_fails_loudly(h5, tmp_path):
    fn = str(tmp_path / "c.h5")
    v = np.arange(200_000, dtype=np.int32)
    h5.write_h5(fn, {"v": v}, blosc=3)
    raw = bytearray(open(fn, "rb").read())
    lo = len(raw) // 2
    raw[lo:lo + 4096] = bytes(4096)                                         # zero a stretch of the compressed chunk
    open(fn, "wb").write(raw)
    with pytest.raises(h5.H5Error):
        h5.read_h5(fn)


@pytest.mark.skipif(not os.path.exists(H5PY_PYTHON), reason="no h5py interpreter in this image")
def test_interchange_with_h5py(h5, tmp_path):
    """Both directions against h5py on the image's libhdf5: files written here read there, and h5py's blosc (blosclz-3
    as Julia's `blosc = 3`, lz4-5), gzip and contiguous datasets read here."""
    d = _sample(np.random.default_rng(2))
    mine, theirs, report = str(tmp_path / "mine.h5"), str(tmp_path / "theirs.h5"), str(tmp_path / "report.json")
    h5.write_h5(mine, d, blosc=3)
    code = f'''
import h5py, json, numpy as np
out = {{}}
with h5py.File({mine!r}) as f:
    for k in f:
        a = f[k][:] if f[k].size else np.zeros(0)
        out[k] = [str(f[k].dtype), list(f[k].shape), {{str(i): list(map(int, v)) for i, v in f[k]._filters.items()}}, float(np.asarray(a, np.float64).sum())]
json.dump(out, open({report!r}, "w"))
rng = np.random.default_rng(3)
with h5py.File({theirs!r}, "w") as f:
    f.create_dataset("userid", data=np.repeat(np.arange(700, dtype=np.int32), 90), chunks=(8192,), compression=32001, compression_opts=(0, 0, 0, 0, 3, 1, 0))
    f.create_dataset("time", data=np.cumsum(rng.random(63000)), chunks=True, compression=32001, compression_opts=(0, 0, 0, 0, 5, 1, 1))
    f.create_dataset("metadata", data=np.arange(70 * 100, dtype=np.float32).reshape(70, 100), compression="gzip")
    f.create_dataset("plain", data=np.arange(12, dtype=np.int64))
'''
    env = dict(os.environ, HDF5_PLUGIN_PATH=h5.PLUGIN_DIR)
    r = subprocess.run([H5PY_PYTHON, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rep = json.load(open(report))
    for k, v in d.items():
        dt, shape, filters, total = rep[k]
        assert dt == str(v.dtype) and tuple(shape) == v.shape, k
        assert total == pytest.approx(float(v.astype(np.float64).sum()), rel=1e-12), k
        if v.size:   # revision 2, blosc format 2, element size, chunk bytes, level 3, byte shuffle, blosclz
            cd = filters["32001"]
            assert cd[:3] == [2, 2, v.dtype.itemsize] and cd[4:] == [3, 1, 0], (k, cd)
    got = h5.read_h5(theirs)
    rng = np.random.default_rng(3)
    assert np.array_equal(got["userid"], np.repeat(np.arange(700, dtype=np.int32), 90))
    assert np.array_equal(got["time"], np.cumsum(rng.random(63000)))
    assert np.array_equal(got["metadata"], np.arange(7000, dtype=np.float32).reshape(70, 100))
    assert np.array_equal(got["plain"], np.arange(12, dtype=np.int64))
    with h5.File(theirs) as f:
        assert [f.info(k)[2] for k in ("userid", "time", "metadata", "plain")] == [3, 5, -1, -1]


# ---- shard writer ------------------------------------------------------------------------------------------------

def _event(medium, item, ts, status, rating, progress=0.5, hs=None, hr=None):
    return dict(medium=medium, matchedid=item, history_max_ts=ts, status=status, rating=rating, progress=progress,
                history_status=hs, history_rating=hr)


def test_get_data_label_rules():
    """transformer.jl:117-143, one event per rule."""
    from recommendersystem_amd.shards import get_data
    user = {"user": {"gender": None, "source": 2}, "items": [
        _event(0, 5, 10.0, 0, 0),                        # status 0, no history: inferred watch
        _event(1, 7, 11.0, 7, 8, hs=3, hr=0),            # planned (3 <= 5) -> watching (7 > 5): new watch + rating + status
        _event(1, 9, 12.0, 7, 8, hs=6, hr=8),            # already past planned, same rating: status target only
        _event(0, 2, 13.0, 4, 0, hs=None, hr=None),      # planned, unrated: status target only
        _event(0, 3, 14.0, 6, 6, hs=6, hr=6),            # nothing changed: projected away
        _event(1, 4, 15.0, 6, 3, hs=0, hr=None),         # history_status 0 is not in (0, 5]: no watch; rating + status
    ]}
    d = get_data(user, 42, num_items_0=100)
    assert len(d) == 27 and all(len(v) == 5 for v in d.values())
    assert d["userid"].tolist() == [42] * 5 and d["gender"].tolist() == [0] * 5 and d["source"].tolist() == [2] * 5
    assert d["matchedid"].tolist() == [5, 107, 109, 2, 104]          # anime ids are offset by num_items(manga)
    assert d["time"].dtype == np.float64 and d["time"].tolist() == [10.0, 11.0, 12.0, 13.0, 15.0]
    assert d["token_mask_ids"].tolist() == [0, 1, 0, 0, 1]
    assert d["0.watch.weight"].tolist() == [1, 0, 0, 0, 0] and d["0.watch.position"].tolist() == [5, 0, 0, 0, 0]
    assert d["1.watch.weight"].tolist() == [0, 1, 0, 0, 0] and d["1.watch.position"].tolist() == [0, 7, 0, 0, 0]
    assert d["1.rating.label"].tolist() == [0, 8, 0, 0, 3] and d["1.rating.weight"].tolist() == [0, 1, 0, 0, 1]
    assert d["0.rating.weight"].tolist() == [0] * 5
    assert d["1.status.label"].tolist() == [0, 7, 7, 0, 6] and d["1.status.position"].tolist() == [0, 7, 9, 0, 4]
    assert d["0.status.label"].tolist() == [0, 0, 0, 4, 0] and d["0.status.weight"].tolist() == [0, 0, 0, 1, 0]
    assert {k: v.dtype for k, v in d.items() if k.endswith("position")}.popitem()[1] == np.int32
    u2 = {"user": {"gender": 1, "source": 0}, "items": [_event(0, 1, 1.0, 7, 0), _event(0, 1, 2.0, 7, 9)]}
    d2 = get_data(u2, 1, 100)                              # a run on one item is one token with the last state
    assert len(d2["userid"]) == 1 and d2["gender"].tolist() == [2] and d2["0.rating.label"].tolist() == [9]
    assert d2["time"].tolist() == [1.0]


def test_rating_rule_compares_in_one_precision_and_finetune_keep_zero():
    """a rating that float32 cannot represent (8.3) equal to its history rating is NOT a rating target
    (transformer.jl:129 compares two Float32 values); and a finetune row with room for the test event only keeps no history."""
    from recommendersystem_amd.shards import NUM_TEST_ITEMS, get_data, get_finetune_data
    user = {"user": {"gender": 0, "source": 1}, "items": [
        _event(1, 9, 12.0, 7, 8.3, hs=6, hr=8.3),       # same non-dyadic rating: status target only
        _event(1, 11, 13.0, 7, 8.3, hs=7, hr=8.2),      # changed rating: rating target
        _event(0, 4, 14.0, 6, 7.1, hs=6, hr=7.1),       # nothing changed: projected away
    ]}
    d = get_data(user, 7, num_items_0=100)
    assert d["matchedid"].tolist() == [109, 111]
    assert d["1.rating.weight"].tolist() == [0, 1] and d["token_mask_ids"].tolist() == [0, 1]
    assert d["1.status.weight"].tolist() == [1, 0]
    row = {"user": {"gender": None, "source": 0}, "items": [_event(0, 1, 1.0, 7, 5), _event(0, 2, 2.0, 7, 6)],
           "test_items": [_event(1, 3, 3.0, 7, 9)]}
    f = get_finetune_data(row, 3, 100, max_seq_len=NUM_TEST_ITEMS)
    assert len(f["userid"]) == NUM_TEST_ITEMS and f["matchedid"].tolist() == [103]
    f3 = get_finetune_data(row, 3, 100, max_seq_len=3)
    assert f3["matchedid"].tolist() == [1, 2, 103]


def test_optdate_and_media_table():
    from recommendersystem_amd.shards import MIN_TS, max_ts_of, media_embedding_matrix, optdate
    max_ts = max_ts_of("20250101\n")
    assert MIN_TS == 946684800.0
    assert optdate(None, max_ts) == (0, 0.0) and optdate("", max_ts) == (0, 0.0)
    assert optdate("2000-01-01", max_ts) == (1, 0.0) and optdate("2025-01-01", max_ts) == (1, 1.0)
    assert optdate("2012-07", max_ts) == optdate("2012-07-01", max_ts)
    assert optdate("2012-00-00", max_ts) == optdate("2012-01-01", max_ts)       # unparseable fields are dropped
    assert optdate("1800-01-01", max_ts) == (1, -5.0)                           # clamp
    assert optdate("soon", max_ts) == (0, 0.0)
    rec = lambda i, sd: {"matchedid": i, "text_embedding": {"embedding": [i + 1.0] * 3}, "image_embedding": [0.5] * 2,
                         "metadata": {"dates": {"startdate": sd, "enddate": None}}}
    W = media_embedding_matrix({0: [rec(1, "2025-01-01")], 1: [rec(0, None)]}, {0: 2, 1: 2}, max_ts, text_dim=3, image_dim=2)
    assert W.shape == (4, 9) and W.dtype == np.float32
    assert W[1].tolist() == [2, 2, 2, 0.5, 0.5, 1, 1, 0, 0] and W[2].tolist() == [1, 1, 1, 0.5, 0.5, 0, 0, 0, 0]
    assert not W[0].any() and not W[3].any()


def test_save_data_to_training_batches(h5, tmp_path):
    """users -> shards (transformer.jl:202-240) -> PretrainDataset (train.py:37-98): equal token counts per shard,
    whole batches, every user's tokens contiguous, and the embeddings file readable by the model-side loader."""
    import msgpack

    from recommendersystem_amd import data, shards
    rng = np.random.default_rng(5)
    datadir = str(tmp_path)
    n_users, lengths = 13, {}
    for u in range(n_users):
        os.makedirs(f"{datadir}/users/training/{u % 3}", exist_ok=True)
        n = int(rng.integers(1, 40))
        items = [_event(int(rng.integers(2)), 1000 * u + j, 100.0 * u + j, int(rng.integers(1, 8)), int(rng.integers(0, 11)))
                 for j in range(n)]                                   # distinct items, no history: nothing is projected away
        lengths[u] = n
        with open(f"{datadir}/users/training/{u % 3}/{u}.msgpack", "wb") as f:
            f.write(msgpack.packb({"user": {"gender": int(u % 2), "source": 1}, "items": items}))
    total = shards.save_data(datadir, "training", "transformer", num_items_0=50_000, num_shards=4, batch_size=64,
                             users_per_part=2, seed=9)
    split = f"{datadir}/transformer/training"
    assert int(open(f"{split}/num_tokens.txt").read()) == total
    counts = [shards.get_num_tokens(split, s) for s in range(1, 5)]
    assert len(set(counts)) == 1 and sum(counts) == total and counts[0] % 64 == 0
    assert any(os.path.exists(f"{split}/{s}/pad.h5") for s in range(1, 5))
    with h5.File(f"{split}/1/1.h5") as f:
        assert len(f.keys()) == 27 and all(f.info(k)[2] == 3 for k in f)
        assert f.info("time")[0] == np.float64 and f.info("userid")[0] == np.int32 and f.info("rating")[0] == np.float32
    seen = 0
    for rank in range(2):
        ds = data.PretrainDataset(split, rank, 2, 64, seed=rank)
        assert len(ds.fns) >= 4
        for batch in ds:
            assert len(batch) == 27 and all(len(v) == 64 for v in batch.values())
            real = batch["time"] > 0
            seen += int(real.sum())
    # 16 users (13 + 3 repeats to fill 4 shards) plus the head of 1.h5 repeated into each pad.h5
    assert seen >= sum(lengths.values())
    table = np.random.default_rng(6).standard_normal((30, 12)).astype(np.float32)
    h5.write_h5(f"{datadir}/media_embeddings.h5", {"metadata": table}, blosc=3)
    with h5.File(f"{datadir}/media_embeddings.h5") as f:
        assert np.array_equal(f["metadata"], table)


def test_finetune_rows_and_dataset(h5, tmp_path):
    """Finetune/transformer.jl:52-166 -> FinetuneDataset (train.py:101-160): one user per row, history then the held-out
    event, targets on that event only, weights normalised, history clipped to the newest max_seq_len - 1 tokens."""
    import msgpack

    from recommendersystem_amd import data, shards
    hist = [_event(0, 10 + j, float(j + 1), 6, 7) for j in range(20)]           # 20 distinct manga items, all kept
    test = _event(1, 3, 99.0, 7, 9, hs=2, hr=None)                               # planned -> watching, first rating
    d = shards.get_finetune_data({"user": {"gender": 0, "source": 3}, "items": hist, "test_items": [test]}, 5, 100, max_seq_len=8)
    assert all(v.shape == (8,) for v in d.values()) and len(d) == 27
    assert d["matchedid"].tolist() == [23, 24, 25, 26, 27, 28, 29, 103]         # newest 7 history tokens, then the test item (anime offset)
    assert d["userid"].tolist() == [5] * 8 and d["gender"].tolist() == [1] * 8
    assert d["1.watch.weight"].tolist() == [0] * 7 + [1] and d["1.watch.position"][7] == 3
    assert d["1.rating.label"][7] == 9 and d["1.rating.weight"][7] == 1 and d["token_mask_ids"].tolist() == [0] * 7 + [1]
    assert d["1.status.label"][7] == 7 and not d["0.rating.weight"].any() and not d["0.watch.weight"].any()
    short = shards.get_finetune_data({"user": {"gender": None, "source": 0}, "items": hist[:2], "test_items": []}, 1, 100, max_seq_len=8)
    assert short["matchedid"].tolist() == [10, 11, 0, 0, 0, 0, 0, 0] and not short["1.watch.weight"].any()
    datadir = str(tmp_path)
    for u in range(5):
        os.makedirs(f"{datadir}/users/training/0", exist_ok=True)
        t = _event(u % 2, 7, 50.0, 7, 8 if u != 3 else 0, hs=1, hr=None)          # user 3: no rating target, still a watch target
        with open(f"{datadir}/users/training/0/{u}.msgpack", "wb") as f:
            f.write(msgpack.packb({"user": {"gender": 1, "source": 1}, "items": hist[:3 + u], "test_items": [t]}))
    n = shards.save_finetune_data(datadir, "training", 100, max_seq_len=16, seed=3)
    assert n == 5
    with h5.File(f"{datadir}/transformer/training/1/1.h5") as f:
        assert f.info("userid")[1] == (5, 16) and f.info("time")[0] == np.float64 and f.info("matchedid")[2] == 3
    ds = data.FinetuneDataset(f"{datadir}/transformer/training", 0, 1, 2, False, 1)
    rows = sum(len(b["userid"]) for b in ds)
    assert rows == 2            # medium-1 users with a watch or rating target: users 1 and 3


def test_event_pipeline_matches_the_reference_on_1200_users():
    """tests/golden/tokenize_1k.npz (oracle/gen_tokenize_fixture.py): the reference's own tokenize / project
    (Finetune/embed.py:39-71 = history_tools.jl:37-75) on 1 200 synthetic users with the importer's history annotation
    (`nothing` on an item's first event).  The shard writer's event pipeline -- shards.tokenize / shards.project and the
    pass-through columns of shards.get_data (matchedid with the anime offset, time, status, rating, progress, user columns)
    -- must reproduce every kept token.  (The target rules of get_data, transformer.jl:120-139, exist only in Julia and
    stay hand-derived: see test_get_data_label_rules.)"""
    from recommendersystem_amd import shards
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "tokenize_1k.npz"))
    NONE = -99
    fields = ("medium", "matchedid", "history_max_ts", "status", "rating", "progress", "history_status", "history_rating")

    def users_of(prefix):
        off = z[prefix + "/offsets"]
        cols = {k: z[f"{prefix}/{k}"] for k in fields}
        out = []
        for a, b in zip(off[:-1], off[1:]):
            ev = []
            for i in range(a, b):
                e = {k: (None if cols[k][i] == NONE else (int(cols[k][i]) if cols[k].dtype == np.int32 else float(cols[k][i]))) for k in fields}
                ev.append(e)
            out.append(ev)
        return out

    raw, want = users_of("in"), users_of("out")
    assert len(raw) == 1200 and sum(map(len, raw)) > 40000
    n0 = 300
    for uid, (events, ref) in enumerate(zip(raw, want)):
        got = shards.project(shards.tokenize([dict(e) for e in events]))
        assert got == ref, uid
        d = shards.get_data({"user": {"gender": None if uid % 3 == 0 else uid % 3 - 1, "source": uid % 4}, "items": [dict(e) for e in events]}, uid + 1, n0)
        assert len(d["userid"]) == len(ref)
        if not ref:
            continue
        np.testing.assert_array_equal(d["matchedid"], [e["matchedid"] + (n0 if e["medium"] == 1 else 0) for e in ref])
        np.testing.assert_array_equal(d["status"], [e["status"] for e in ref])
        np.testing.assert_array_equal(d["rating"], np.array([e["rating"] for e in ref], np.float32))
        np.testing.assert_array_equal(d["progress"], np.array([e["progress"] for e in ref], np.float32))
        np.testing.assert_array_equal(d["time"], [e["history_max_ts"] for e in ref])
        assert (d["userid"] == uid + 1).all() and (d["source"] == uid % 4).all() and (d["gender"] == (0 if uid % 3 == 0 else uid % 3)).all()
