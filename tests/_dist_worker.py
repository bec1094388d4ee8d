"""Worker of tests/test_dist_cpu.py: one rank of a world_size-2 host group (TCP control plane) on CPU.  torch is made
un-importable first: a rank of the data-parallel path must never need it."""
import json
import os
import sys

import numpy as np

sys.modules["torch"] = None            # any `import torch` below raises ImportError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import model_np, synth  # noqa: E402
from recommendersystem_amd import dist as rdist  # noqa: E402
from recommendersystem_amd.train import make_task_weights, reduce_mean  # noqa: E402


class HostOnlyComm:
    """Same surface as recommendersystem_amd.dist.Comm, but the reductions run over the TCP control plane on host
    arrays (the RCCL communicator needs GPUs)."""

    def __init__(self, hg):
        self.hg, self.rank, self.world = hg, hg.rank, hg.world

    def all_reduce_sum(self, values):
        return self.hg.all_reduce(values, "sum")


def main():
    out_path = sys.argv[1]
    hg = rdist.HostGroup()
    rank, world = hg.rank, hg.world
    res = {"rank": rank, "world": world}
    # control plane
    payload = bytes(range(128)) if rank == 0 else bytes(128)
    res["bcast_ok"] = hg.broadcast_bytes(payload, 0) == bytes(range(128))
    res["sum"] = hg.all_reduce([rank + 1.0, 10.0 * (rank + 1)], "sum")
    res["max"] = hg.all_reduce([float(rank)], "max")
    hg.barrier()
    # data path semantics (DDP mean of per-rank gradients, train.py:678-682): rank r owns stream seed^r
    cfg = synth.make_config("tiny", mask_rate=0.25, mask_topk=6)
    P = synth.make_params(cfg, 3, "test")
    rows = 2
    d = synth.make_batch(cfg, rows, 100 ^ rank)
    wm, rm = synth.make_masks(cfg, rows, 200 ^ rank)
    tw = make_task_weights()
    model = model_np.OracleModel(cfg, P)
    dm = model_np.mask_tokens(cfg, model_np.reshape_batch(cfg, d), wm, rm)
    losses, G = model.forward(dm, False, True, tw)
    names = synth.trainable_names(cfg)
    flat = np.concatenate([G[n].reshape(-1) for n in names])
    red = np.array(hg.all_reduce(flat.tolist(), "sum")) / world
    res["grad_mean_norm"] = float(np.sqrt((red ** 2).sum()))
    res["grad_local_norm"] = float(np.sqrt((flat ** 2).sum()))
    # epoch metrics (reduce_mean, train.py:199-204)
    wsums = [float(dm[f"{m}.{k}.weight"].sum()) for m in (0, 1) for k in ("watch", "rating")]
    res["reduce_mean"] = reduce_mean(HostOnlyComm(hg), [l * w for l, w in zip(losses, wsums)], wsums)
    res["losses"] = losses
    res["wsums"] = wsums
    shards = [f"s{i}" for i in range(8)]
    res["shards"] = rdist.shard_for_rank(shards, rank, world)
    arr = np.full(1000, rank + 1.0, np.float32)
    res["array_sum_ok"] = bool((hg.all_reduce_array(arr) == sum(range(1, world + 1))).all())
    res["min"] = hg.all_reduce([float(rank)], "min")
    res["gathered"] = [len(b) for b in hg.all_gather_bytes(bytes([rank]) * (3 + 2 * rank))] + [hg.all_gather_bytes(b"ab" if rank else b"")[0] == b""]
    res["torch_blocked"] = "torch" in sys.modules and sys.modules["torch"] is None
    np.save(out_path + f".grad{rank}.npy", flat)
    with open(out_path + f".{rank}.json", "w") as f:
        json.dump(res, f)
    hg.close()


if __name__ == "__main__":
    main()
