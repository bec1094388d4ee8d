"""Data-parallel plumbing: one process per GPU (train.py:582-587), gradient all-reduce over RCCL/xGMI.

The data path (gradient buckets) goes through the library's own RCCL communicator
(rsys_comm_*, include/rsys.h).  The control plane -- exchanging the 128-byte RCCL id,
barriers, the scalar max/sum of the benchmark and of reduce_mean -- uses the
rendezvous the launcher already provides (torchrun env: RANK, WORLD_SIZE, LOCAL_RANK,
MASTER_ADDR, MASTER_PORT) through torch.distributed's gloo backend on CPU tensors.
"""
import ctypes as C
import os

import numpy as np

from ._lib import check, lib


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


class HostGroup:
    """CPU-side process group (gloo).  world == 1 needs no torch at all."""

    def __init__(self, rank=None, world=None):
        r, w, _ = env_rank()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.pg = None
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29500")
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self.pg = dist

    def barrier(self):
        if self.pg is not None:
            self.pg.barrier()

    def broadcast_bytes(self, data, src=0):
        if self.pg is None:
            return data
        import torch
        t = torch.tensor(list(data), dtype=torch.uint8) if self.rank == src else torch.zeros(len(data), dtype=torch.uint8)
        self.pg.broadcast(t, src)
        return bytes(t.tolist())

    def all_reduce(self, values, op="sum"):
        if self.pg is None:
            return [float(v) for v in values]
        import torch
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        self.pg.all_reduce(t, op=self.pg.ReduceOp.SUM if op == "sum" else self.pg.ReduceOp.MAX)
        return t.tolist()

    def close(self):
        if self.pg is not None and self.pg.is_initialized():
            self.pg.destroy_process_group()
            self.pg = None


class Comm:
    """RCCL communicator of this rank (ncclCommInitRank) + the DDP-style gradient all-reduce."""

    def __init__(self, host_group, device):
        self.hg = host_group
        self.rank, self.world = host_group.rank, host_group.world
        self._h = C.c_void_p()
        ident = (C.c_uint8 * 128)()
        if self.rank == 0:
            check(lib().rsys_comm_unique_id(C.byref(ident)))
        raw = host_group.broadcast_bytes(bytes(ident), 0)
        ident = (C.c_uint8 * 128)(*raw)
        check(lib().rsys_comm_init(C.byref(ident), self.rank, self.world, device, C.byref(self._h)))

    def self_test(self):
        """hardware_check.py:6-12: all-reduce of ones must equal the world size."""
        check(lib().rsys_self_test(self._h))

    def begin_grad_sync(self, model):
        """Before the backward of an optimizer step's last micro-step: finished gradient buckets of the trunk are reduced
        while the backward is still running (DDP's bucket hooks, train.py:678-682)."""
        check(lib().rsys_set_grad_sync(model._h, self._h))

    def all_reduce_grads(self, model):
        check(lib().rsys_allreduce_grads(model._h, self._h))

    def early_reduced(self, model):
        """gradient elements the last all_reduce_grads found already reduced by the backward's bucket hooks"""
        n = C.c_int64()
        check(lib().rsys_grad_sync_early(model._h, C.byref(n)))
        return n.value

    def all_reduce_sum(self, values):
        arr = (C.c_double * len(values))(*[float(v) for v in values])
        check(lib().rsys_allreduce_f64(self._h, arr, len(values)))
        return list(arr)

    def close(self):
        if self._h:
            lib().rsys_comm_destroy(self._h)
            self._h = C.c_void_p()


class HostComm:
    """Gradient all-reduce through host memory over gloo (same surface as Comm): a debugging aid for machines whose RCCL
    cannot initialise.  Never chosen silently: `make_comm` raises unless RSYS_ALLOW_HOST_ALLREDUCE=1 is set, and the
    class says what it is on stderr (it is orders of magnitude slower than RCCL over xGMI)."""

    def __init__(self, host_group, reason=""):
        import sys
        self.hg = host_group
        self.rank, self.world = host_group.rank, host_group.world
        if self.rank == 0:
            print(f"[recommendersystem_amd.dist] RCCL communicator unavailable ({reason}); "
                  "falling back to a HOST all-reduce over gloo", file=sys.stderr)

    def self_test(self):
        assert self.hg.all_reduce([1.0], "sum")[0] == float(self.world)

    def begin_grad_sync(self, model):
        pass

    def all_reduce_grads(self, model):
        import torch
        ptr = C.c_void_p(); n = C.c_int64()
        check(lib().rsys_grad_buffer(model._h, C.byref(ptr), C.byref(n)))
        host = np.empty(n.value, np.float32)
        check(lib().rsys_dev_d2h(host.ctypes.data, ptr, host.nbytes))
        t = torch.from_numpy(host)
        self.hg.pg.all_reduce(t)
        check(lib().rsys_dev_h2d(ptr, host.ctypes.data, host.nbytes))

    def all_reduce_sum(self, values):
        return self.hg.all_reduce(values, "sum")

    def close(self):
        pass


def _no_rccl(host_group, reason):
    if os.environ.get("RSYS_ALLOW_HOST_ALLREDUCE") == "1":
        return HostComm(host_group, reason)
    raise RuntimeError(f"RCCL communicator unavailable on at least one rank ({reason}); the gradient all-reduce has no "
                       "silent host path (set RSYS_ALLOW_HOST_ALLREDUCE=1 to debug with a gloo all-reduce)")


def make_comm(host_group, device):
    """RCCL communicator of this rank; every rank takes the same branch (the outcome of the RCCL attempt is agreed on
    with a MIN reduction over gloo, so a failure raises on all ranks instead of leaving some inside a collective)."""
    if host_group.world == 1:
        return None
    comm, err = None, ""
    # pre-flight on every rank (RCCL loadable, id can be made) BEFORE any collective RCCL call: a rank that cannot load
    # the library must not leave the others waiting inside ncclCommInitRank
    try:
        scratch = (C.c_uint8 * 128)()
        check(lib().rsys_comm_unique_id(C.byref(scratch)))
        loadable = 1.0
    except Exception as e:   # noqa: BLE001
        loadable, err = 0.0, str(e)
    if -host_group.all_reduce([-loadable], "max")[0] < 1.0:
        return _no_rccl(host_group, err or "RCCL not loadable on another rank")
    try:
        comm = Comm(host_group, device)
        comm.self_test()
    except Exception as e:   # noqa: BLE001 - any failure of the native path selects the fallback
        err = str(e)
        comm = None
    ok_everywhere = -host_group.all_reduce([-(1.0 if comm is not None else 0.0)], "max")[0]
    if ok_everywhere >= 1.0:
        return comm
    if comm is not None:
        comm.close()
    return _no_rccl(host_group, err or "another rank failed")


def shard_for_rank(shards, local_rank, local_world_size):
    """train.py:46-51: shard directory i goes to rank i % world; the count must divide evenly."""
    assert len(shards) % local_world_size == 0
    return [x for i, x in enumerate(shards) if i % local_world_size == local_rank]
