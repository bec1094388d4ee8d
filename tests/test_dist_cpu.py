"""CPU, world_size 2 over gloo: the N>1 host path (rendezvous from the launcher's env, id broadcast,
scalar reductions, DDP-mean semantics of the gradient all-reduce, reduce_mean, shard assignment)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_world2_gloo(tmp_path):
    port = _free_port()
    out = str(tmp_path / "res")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), out], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    res = [json.load(open(out + f".{r}.json")) for r in range(2)]
    g = [np.load(out + f".grad{r}.npy") for r in range(2)]
    for r in res:
        assert r["bcast_ok"]
        assert r["sum"] == [3.0, 30.0] and r["max"] == [1.0]
    mean = (g[0] + g[1]) / 2
    assert abs(res[0]["grad_mean_norm"] - np.sqrt((mean ** 2).sum())) < 1e-9
    assert res[0]["grad_mean_norm"] == res[1]["grad_mean_norm"]          # every rank holds the same averaged gradient
    # reduce_mean == pooled weighted mean over both ranks
    for i in range(4):
        num = sum(r["losses"][i] * r["wsums"][i] for r in res); den = sum(r["wsums"][i] for r in res)
        exp = num / den if den else 0
        assert abs(res[0]["reduce_mean"][i] - exp) < 1e-9 and res[0]["reduce_mean"] == res[1]["reduce_mean"]
    assert res[0]["shards"] == ["s0", "s2", "s4", "s6"] and res[1]["shards"] == ["s1", "s3", "s5", "s7"]
