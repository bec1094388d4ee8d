"""Process seam of the training path: the command line of notebooks/Training/transformer.py.

    torchrun --standalone --nproc_per_node=N -m recommendersystem_amd.cli --datadir DIR [--mini] [--prod]
    python -m recommendersystem_amd.cli --datadir DIR --finetune CKPT --finetune_medium {0,1} --finetune_metric {watch,rating}

mirrors `train()` (transformer.py:583-757): one process per GPU from the launcher's RANK / LOCAL_RANK / WORLD_SIZE,
configuration from `{manga,anime}.csv` + `list_tag` (transformer.py:513-567), shards under `transformer[_mini]/{training,test}`,
`media_embeddings.h5`, resume from `transformer.masked.npz` under --prod (converted from / to the reference's `.pt` by
recommendersystem_amd.checkpoint), WSD schedule from `num_tokens.txt`, the epoch loop with evaluation, early stopping,
checkpoint + metrics CSV, and `transformer.masked.finished` at the end of a --prod run.  Not here: `--download` and the
R2 upload (rclone; SURVEY marks storage out of scope).

Differences that are hardware choices, not behaviour: the local batch size defaults to 64 rows (the reference's value for
its largest GPU; --local_batch_size overrides), arithmetic is bf16 with fp32 accumulation (--dtype fp32 for the parity
mode; the reference's production run uses fp8 linears), model sizes other than the production one are selectable
(--model cfg3 ...) because the synthetic benchmarks use them.
"""
import argparse
import csv
import datetime
import os
import sys

import numpy as np

from . import data as rdata
from . import dist as rdist
from . import train as rtrain
from . import workload
from .model import RecommenderModel
from .optim import create_optimizer

PROD_DIMS = dict(num_layers=8, num_heads=32, num_kv_heads=16, embed_dim=2048, intermediate_dim=5632, max_sequence_length=1024,
                 mask_topk=128)                                                                   # transformer.py:536-558


def get_num_items(datadir, medium, col="matchedid"):
    """transformer.py:528-530: largest id in `{medium}.csv` + 1."""
    best = -1
    with open(os.path.join(datadir, f"{medium}.csv"), newline="") as f:
        for row in csv.DictReader(f):
            v = row.get(col, "")
            if v not in ("", None):
                best = max(best, int(float(v)))
    assert best >= 0, f"{medium}.csv has no {col} column"
    return best + 1


def get_training_config(args):
    """transformer.py:513-567."""
    if args.finetune is not None:
        config = load_checkpoint_blob(args.finetune)[1]
        config["learning_rate"] = 2e-4
        config["finetune"] = True
        config["finetune_metric"] = args.finetune_metric
        return config
    min_ts = datetime.datetime.strptime("20000101", "%Y%m%d").timestamp()
    with open(os.path.join(args.datadir, "list_tag")) as f:
        max_ts = datetime.datetime.strptime(f.read().strip(), "%Y%m%d").timestamp()
    dims = dict(PROD_DIMS)
    if args.model != "prod":
        c = workload.make_config(args.model)
        dims = {k: c[k] for k in PROD_DIMS}
    config = dict(dims)
    config.update({
        "vocab_sizes": {"0_matchedid": get_num_items(args.datadir, "manga"), "1_matchedid": get_num_items(args.datadir, "anime"),
                        "status": 9, "gender": 4, "source": 4},
        "metadata_emb_size": args.metadata_emb_size,
        "min_ts": min_ts, "max_ts": max_ts, "rating_mean": 7.6287384, "rating_std": 1.778219,
        "forward": "train", "finetune": False, "learning_rate": 1e-4, "mask_rate": 0.1,
    })
    assert config["mask_topk"] > config["mask_rate"] * config["max_sequence_length"]
    if args.mini:
        assert config["num_layers"] % 2 == 0
        config["num_layers"] //= 2
    return config


def load_checkpoint_blob(path):
    """(flat npz-layout dict, config) of a checkpoint given as this package's `.npz` or the reference's `.pt`."""
    import json
    if path.endswith(".pt"):
        import torch

        from .checkpoint import from_reference
        blob = from_reference(torch.load(path, weights_only=False, map_location="cpu"))
    else:
        z = np.load(path)
        blob = {k: z[k] for k in z.files}
    return blob, json.loads(bytes(np.asarray(blob["config"], np.uint8)).decode())


def main(argv=None):
    ap = argparse.ArgumentParser(prog="recommendersystem_amd.cli")
    ap.add_argument("--datadir", type=str, required=True)
    ap.add_argument("--finetune", type=str, default=None)
    ap.add_argument("--finetune_medium", type=int, default=None)
    ap.add_argument("--finetune_metric", type=str, default=None)
    ap.add_argument("--mini", action="store_true")
    ap.add_argument("--prod", action="store_true")
    ap.add_argument("--model", default="prod", help="prod (transformer.py:536-558) or a workload.make_config name")
    ap.add_argument("--metadata_emb_size", type=int, default=3072 * 2 + 4)
    ap.add_argument("--dtype", default="bf16", choices=("bf16", "fp32", "fp8"))
    ap.add_argument("--local_batch_size", type=int, default=None)
    ap.add_argument("--global_batch_size", type=int, default=None, help="rows per optimizer step (transformer.py:593, 600: 32 / 512)")
    ap.add_argument("--num_epochs", type=int, default=None)
    ap.add_argument("--warmup_steps", type=int, default=2000, help="transformer.py:345 (fixed there; short rehearsal runs need fewer)")
    ap.add_argument("--table_shard", action="store_true",
                    help="row-sharded item table + vocabulary-parallel cross entropy over the ranks (SURVEY 8(e) cfg-4; beyond the reference)")
    ap.add_argument("--sampled_softmax", type=int, default=0, help="with --table_shard: classes sampled per rank and medium (0 = full soft-max)")
    ap.add_argument("--nproc_per_node", type=int, default=None,
                    help="start this many ranks of this command (what `torchrun --standalone --nproc_per_node=N` does in "
                         "entrypoint.sh:25); ignored when a launcher has already set WORLD_SIZE")
    args = ap.parse_args(argv)
    if args.nproc_per_node and args.nproc_per_node > 1 and "WORLD_SIZE" not in os.environ:
        # this process has made no GPU call: spawn the ranks as ordinary children and pass their verdict on
        raw = list(sys.argv[1:] if argv is None else argv)
        sys.exit(rdist.launch_local(args.nproc_per_node, [sys.executable, "-m", "recommendersystem_amd.cli"] + raw))

    rank, world, local_rank = rdist.env_rank()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    log = (lambda *a: print(*a, file=sys.stderr, flush=True)) if local_rank == 0 else (lambda *a: None)
    config = get_training_config(args)
    run = rtrain.get_run_config(bool(config["finetune"]))
    local_batch = args.local_batch_size or run["local_batch_size"]
    if args.global_batch_size:
        run["global_batch_size"] = args.global_batch_size
    num_epochs = args.num_epochs or run["num_epochs"]
    if config["finetune"]:
        assert world == 1, "finetuning runs on one GPU (transformer.py:591-597)"
        assert run["global_batch_size"] % local_batch == 0
        grad_accum_steps = run["global_batch_size"] // local_batch
    else:
        assert run["global_batch_size"] % (world * local_batch) == 0
        grad_accum_steps = run["global_batch_size"] // (world * local_batch)
    config["local_batch_size"] = local_batch

    transdir = "transformer_mini" if args.mini else "transformer"

    def dataset(split):
        path = f"{args.datadir}/{transdir}/{split}"
        if config["finetune"]:
            return rdata.FinetuneDataset(path, local_rank, local_world, local_batch, split == "training", args.finetune_medium)
        return rdata.PretrainDataset(path, local_rank, local_world, local_batch * config["max_sequence_length"], seed=rank)

    dataloaders = {x: rdata.Prefetch(dataset(x)) for x in ("training", "test")}   # (the DataLoader workers of train.py:162-165)
    hg = rdist.HostGroup()
    if args.table_shard:
        assert not config["finetune"], "finetuning keeps the (frozen) item table replicated"
        config["table_shard"] = (rank, world)
        if args.sampled_softmax:
            config["sampled_softmax"] = args.sampled_softmax
    model = RecommenderModel(config, device=local_rank, dtype=args.dtype, max_rows=local_batch)
    model.load_pretrained_embeddings(args.datadir)
    checkpoint_fn = f"{args.datadir}/transformer.masked.npz"
    resume = None
    if config["finetune"]:
        blob, _ = load_checkpoint_blob(args.finetune)
        model.load_state_dict({k[len("model/"):]: v for k, v in blob.items() if k.startswith("model/")}, strict=False)
    elif os.path.exists(checkpoint_fn) and args.prod:
        resume = checkpoint_fn
    else:
        model.init_weights(0x1217)                     # the same seed on every rank replaces DDP's rank-0 broadcast
    n_all = sum(int(np.prod(s)) for _, s, _ in model.named_parameters())
    n_train = sum(int(np.prod(s)) for _, s, t in model.named_parameters() if t)
    log(f"Created model with {n_all} parameters and {n_train} trainable parameters")
    comm = rdist.make_comm(hg, local_rank)
    if args.table_shard and comm is not None:
        model.set_shard_comm(comm)
    optimizer = create_optimizer(model, config)
    if config["finetune"]:
        scheduler = rtrain.create_learning_rate_schedule(0, 1, num_epochs, finetune=True)
    else:
        with open(f"{args.datadir}/{transdir}/training/num_tokens.txt") as f:
            tokens_per_epoch = int(f.read().strip())
        scheduler = rtrain.create_learning_rate_schedule(tokens_per_epoch, run["global_batch_size"] * config["max_sequence_length"], num_epochs,
                                                          warmup_steps=args.warmup_steps)
        log(f"Training with {tokens_per_epoch * num_epochs} tokens and {tokens_per_epoch} tokens per epoch")
    starting_epoch = 0
    if resume is not None:
        epoch, _ = rtrain.load_checkpoint(resume, model, optimizer, scheduler)
        log(f"loading model and optimizer state from epoch {epoch}")
        starting_epoch = epoch + 1
    task_weights = rtrain.make_task_weights(args.finetune_medium, args.finetune_metric) if config["finetune"] else rtrain.make_task_weights()
    basename = "transformer.masked" if not config["finetune"] else f"transformer.masked.{args.finetune_medium}.{args.finetune_metric}.finetune"
    history = rtrain.train(model, optimizer, scheduler, dataloaders, config, args.datadir, task_weights, num_epochs, grad_accum_steps,
                           comm, rank, starting_epoch, basename, log, gather=hg if (args.table_shard and world > 1) else None)
    if comm is not None:
        comm.close()
    hg.close()
    if rank == 0 and args.prod and not config["finetune"]:
        open(f"{args.datadir}/transformer.masked.finished", "w").close()         # transformer.py:507-508 (the upload itself is out of scope)
    model.close()
    return history


if __name__ == "__main__":
    main()
