"""Rehearsal of the multi-rank row-sharded path (cfg-4 machinery) at REAL size on one GPU: `world` concurrent ranks of the in-process
rank group, each holding 1/world of the item tables of a cfg-3 / cfg-4 model, take optimizer steps on their own batches (sparse row
exchange, vocabulary-parallel or sampled soft-max heads, dense all-reduce of the replicated parameters, global-norm clip, AdamW);
beside them the same ranks with a REPLICATED table (the data-parallel scheme that is pinned to the oracle).  Full soft-max:
the losses of the two schemes must agree step by step."""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recommendersystem_amd as ra  # noqa: E402
from recommendersystem_amd import dist as rdist, workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cfg3")
ap.add_argument("--world", type=int, default=2)
ap.add_argument("--rows", type=int, default=32)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--sampled", type=int, default=0)
args = ap.parse_args()
cfg = workload.make_config(args.config)
W, rows = args.world, args.rows
batches = [workload.make_batch(cfg, rows, 0xD47A ^ r, mu=4.6, sigma=1.0) for r in range(W)]
tw = ra.make_task_weights()


def run(sharded, sampled):
    group = rdist.LocalGroup(W)
    out = [None] * W; err = [None] * W

    def rank(r):
        try:
            comm = rdist.LocalComm(group, r)
            c = dict(cfg)
            if sharded:
                c["table_shard"] = (r, W)
                if sampled:
                    c["sampled_softmax"] = sampled
            model = ra.RecommenderModel(c, device=0, dtype="bf16", max_rows=rows)
            if sharded:
                model.set_shard_comm(comm)
            model.init_weights(0x1217); model.random_pretrained_embeddings(0x3E7A)
            opt = ra.create_optimizer(model, c)
            model.set_loss_weights(tw, 1)
            model.mask_seed = 0x3A5C ^ r
            model.upload(batches[r])
            ls, t0 = [], None
            for s in range(args.steps):
                if s == 1:
                    ra.synchronize(); t0 = time.perf_counter()
                comm.begin_grad_sync(model)
                model.forward_resident(False)
                comm.all_reduce_grads(model)
                opt.step(lr_factor=1.0, clip_max_norm=1.0, grad_div=float(W))
                ls.append(model.losses(False))
            ra.synchronize()
            dt = (time.perf_counter() - t0) / max(1, args.steps - 1)
            model.close(); comm.close()
            out[r] = (ls, dt)
        except BaseException as e:   # noqa: BLE001
            err[r] = e
    th = [threading.Thread(target=rank, args=(r,)) for r in range(W)]
    for t in th:
        t.start()
    for t in th:
        t.join(900)
    group.close()
    for e in err:
        if e is not None:
            raise e
    return out


rep = run(False, 0)
sh = run(True, args.sampled)
for r in range(W):
    print(f"rank {r}: replicated  {[np.round(l, 4).tolist() for l in rep[r][0]]}  {rep[r][1] * 1e3:.1f} ms/step (all {W} ranks share the GPU)")
    print(f"rank {r}: row-sharded {[np.round(l, 4).tolist() for l in sh[r][0]]}  {sh[r][1] * 1e3:.1f} ms/step" + (f"  (sampled soft-max {args.sampled})" if args.sampled else ""))
if not args.sampled:
    worst = max(abs(a - b) / max(abs(b), 1.0) for r in range(W) for la, lb in zip(sh[r][0], rep[r][0]) for a, b in zip(la, lb))
    print("worst relative loss difference sharded vs replicated:", worst)
    assert worst < 5e-3, worst
print("ok")
