"""BASELINE configs[4] (cfg-5) on one MI355X: the daily finetune loop of notebooks/Finetune/run.jl:9-13 at the benchmark's model
size (cfg-3 dims: D=512, L=8, S=512, 200 K items, M=6148) on HDF5 finetune shards in the reference's layout.

  1. synthetic users -> `users/{training,test}/0/*.msgpack` (history + one held-out test event each, the importer's history
     annotation) -> `shards.save_finetune_data` (Finetune/transformer.jl:135-166: one user per row, blosc HDF5);
  2. a base checkpoint of the pretraining model (random init, `.npz`) and media_embeddings.h5;
  3. the four LoRA runs medium x metric through the command line (`cli.main --finetune BASE --finetune_medium m
     --finetune_metric k`): frozen base, rank-8 updates on q and v, batch 16 x accumulation 2, early stopper.
Prints one JSON object with the time, step rate and losses of every run (profiles/r2_cfg5_finetune.json).

    python tools/run_cfg5.py [--users 1536] [--epochs 3] [--model cfg3]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_user(rng, n_events, V):
    """history in time order with the previous state of the item on every event, then one held-out event"""
    items, ts, snap = [], 1.0e9 + float(rng.integers(0, 10 ** 8)), {}
    for _ in range(n_events + 1):
        m = int(rng.random() < 0.7)
        mid = (m, int(min(V[m] - 1, rng.zipf(1.2))))
        ts += float(rng.integers(60, 400000))
        st = int(rng.integers(1, 9))
        rt = float(rng.integers(1, 11)) if rng.random() > 0.45 else 0.0
        hs, hr = snap.get(mid, (None, None))
        items.append({"medium": mid[0], "matchedid": mid[1], "history_max_ts": ts, "status": st, "rating": rt,
                      "progress": float(rng.random()), "history_status": hs, "history_rating": hr})
        snap[mid] = (st, rt)
    return {"user": {"gender": int(rng.integers(0, 3)), "source": int(rng.integers(0, 4))}, "items": items[:-1], "test_items": items[-1:]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=1536)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--model", default="cfg3")
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    import msgpack

    import recommendersystem_amd as ra
    from recommendersystem_amd import cli, h5, shards, train, workload
    cfg = workload.make_config(args.model)
    V0, V1, S, M = cfg["vocab_sizes"]["0_matchedid"], cfg["vocab_sizes"]["1_matchedid"], cfg["max_sequence_length"], cfg["metadata_emb_size"]
    rng = np.random.default_rng(5)
    out = {"workload": f"cfg-5: 4 LoRA finetune runs (medium x metric) sharing one base, model {args.model} (D={cfg['embed_dim']} L={cfg['num_layers']} "
                       f"S={S} V={V0 + V1} M={M}), {args.users} training users, batch 16 x accum 2, {args.epochs} epochs", "runs": []}
    with tempfile.TemporaryDirectory() as d:
        t0 = time.time()
        open(f"{d}/manga.csv", "w").write("matchedid\n" + "\n".join(str(i) for i in range(V0)) + "\n")
        open(f"{d}/anime.csv", "w").write("matchedid\n" + "\n".join(str(i) for i in range(V1)) + "\n")
        open(f"{d}/list_tag", "w").write("20260101")
        for split, n in (("training", args.users), ("test", max(64, args.users // 8))):
            os.makedirs(f"{d}/users/{split}/0")
            for u in range(n):
                with open(f"{d}/users/{split}/0/{u}.msgpack", "wb") as f:
                    f.write(msgpack.packb(make_user(rng, int(np.clip(rng.lognormal(4.6, 1.0), 5, 2000)), (V0, V1))))
            shards.save_finetune_data(d, split, V0, max_seq_len=S, seed=3)
        base = (rng.standard_normal((997, M)) / np.sqrt(M)).astype(np.float32)
        table = base[np.arange(V0 + V1) % 997] * (1.0 + (np.arange(V0 + V1) % 13)[:, None].astype(np.float32) / 13.0)
        h5.write_h5(f"{d}/media_embeddings.h5", {"metadata": table}, blosc=3)
        del table
        cfg.update({"finetune": False, "forward": "train", "learning_rate": 1e-4})
        model = ra.RecommenderModel(cfg, dtype=args.dtype, max_rows=1)
        model.init_weights(0x1217)
        train.checkpoint_model(d, model, None, None, cfg, 0, [0.0] * 4, [0.0] * 4, train.make_task_weights(), True, basename="base")
        model.close()
        out["prepare_s"] = round(time.time() - t0, 1)
        for medium in (0, 1):
            for metric in ("watch", "rating"):
                t1 = time.time()
                hist = cli.main(["--datadir", d, "--finetune", f"{d}/base.npz", "--finetune_medium", str(medium), "--finetune_metric", metric,
                                 "--model", args.model, "--metadata_emb_size", str(M), "--dtype", args.dtype, "--num_epochs", str(args.epochs)])
                dt = time.time() - t1
                ti = medium * 2 + (0 if metric == "watch" else 1)
                out["runs"].append({"medium": medium, "metric": metric, "epochs_run": len(hist), "wall_s": round(dt, 1),
                                    "train_loss": [round(float(tr[ti]), 4) for _, tr, _ in hist],
                                    "test_loss": [round(float(te[ti]), 4) for _, _, te in hist]})
                assert all(np.isfinite(l).all() for _, tr, te in hist for l in (tr, te))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
