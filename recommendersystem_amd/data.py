"""Shard reader mirroring PretrainDataset (notebooks/Training/transformer.py:37-98).

The reference reads blosc-compressed HDF5 shards written by Training/transformer.jl:202-240 (written here by
shards.py); `.h5` files go through the libhdf5 + blosc adapter (h5.py, include/rsys_h5.h), and an `.npz` with the
same 27 keys and dtypes is accepted beside them (tests and hosts without libhdf5).  Directory layout and rank
assignment are the reference's: `{datadir}/{shard}/{p}.h5`, shard directory i -> rank i % world (train.py:46-51).
"""
import glob
import os

import numpy as np

from .dist import shard_for_rank


def shard_files(shard_dir):
    """train.py:51: every `.h5` of a shard directory (pad.h5 included), plus `.npz` stand-ins."""
    return sorted(glob.glob(f"{shard_dir}/*.h5")) + sorted(glob.glob(f"{shard_dir}/*.npz"))


def load_shard(fn):
    """train.py:86-89: `for k in f: d[k] = f[k][:]`."""
    if fn.endswith(".h5"):
        from . import h5
        return h5.read_h5(fn)
    with np.load(fn) as f:
        return {k: f[k] for k in f.files}


def get_index_permutation(arr, rng):
    """train.py:53-68: the stream is cut where the userid changes, the users' runs are drawn in a random order (one
    `rng.permutation` over the runs, as the reference draws it), positions inside a run keep their order."""
    arr = np.asarray(arr)
    cuts = np.flatnonzero(arr[1:] != arr[:-1]) + 1
    first = np.r_[0, cuts]
    length = np.diff(np.r_[first, len(arr)])
    order = rng.permutation(len(first))
    # gather all runs at once: position i of the output belongs to run order[j], offset i - (start of slot j)
    slot_start = np.cumsum(length[order]) - length[order]
    return (np.repeat(first[order] - slot_start, length[order]) + np.arange(len(arr))).astype(np.int64)


def block_shuffle(d, rng):
    """train.py:70-73."""
    p = get_index_permutation(d["userid"], rng)
    for k in d:
        d[k] = d[k][p]


class Prefetch:
    """The reference feeds its loop through a torch DataLoader with worker processes (train.py:162-165, `num_workers`): shard
    decode and block shuffle overlap the GPU step.  Here one producer thread runs the wrapped dataset's iterator `depth` batches
    ahead (blosc / HDF5 decode and the device calls of the consumer both release the GIL); order and content are the
    dataset's own, an exception in the producer is re-raised in the consumer, leaving the loop early stops the producer."""

    # seconds: interpreter switch interval while the producer thread runs (tools/ab_loader_interval.py).  The interval is a property of the
    # whole interpreter, so it is changed only between the first batch asked for and the end of the iteration (restored on exhaustion, on
    # `close()` of the generator and on an exception alike), and never when the embedding application set RSYS_KEEP_SWITCH_INTERVAL=1 or
    # passes switch_interval=None.
    switch_interval = 1e-4

    def __init__(self, dataset, depth=16, switch_interval="default"):
        if switch_interval != "default":
            self.switch_interval = switch_interval
        # depth: batches the producer may run ahead.  A shard file is decoded and block-shuffled in one go (~60 ms for 8 batches of
        # cfg-3), so a queue of 3 ran dry at every file boundary; 16 = the reference's 8 workers x prefetch_factor 2 (train.py:162-165)
        self.dataset, self.depth = dataset, max(1, int(depth))

    def __getattr__(self, name):          # (`partition` of a FinetuneDataset and the like)
        return getattr(self.dataset, name)

    def __iter__(self):
        import queue
        import threading
        q = queue.Queue(self.depth)
        stop = threading.Event()
        END = object()

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1); return True
                except queue.Full:
                    continue
            return False

        def produce():
            try:
                for item in self.dataset:
                    if not put(item):
                        return
                put(END)
            except BaseException as e:     # noqa: B902 -- handed to the consumer
                put(e)

        # The consumer re-takes the interpreter lock after every device call it returns from; with the default 5 ms switch interval
        # a producer busy in numpy / Python code holds it that long and the GPU waits for its next launches (measured: the HDF5-fed
        # loop +4 % over the in-memory loop, gpurun_out/r5f_bench.json).  A short interval hands the lock over within tens of microseconds (1e-4 s: +2.2 %, gpurun_out/r5g_bench_full.json).
        import os
        import sys
        old_interval = sys.getswitchinterval()
        if self.switch_interval is not None and os.environ.get("RSYS_KEEP_SWITCH_INTERVAL") != "1":
            sys.setswitchinterval(min(old_interval, self.switch_interval))
        t = threading.Thread(target=produce, daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is END:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()
            t.join(timeout=5.0)
            sys.setswitchinterval(old_interval)


class PretrainDataset:
    """Iterates batches of `tokens_per_batch` interactions; users may straddle batch boundaries exactly as in
    the reference (train.py:91-98)."""

    def __init__(self, datadir, local_rank, local_world_size, tokens_per_batch, seed=0):
        self.batch_size = tokens_per_batch
        shards = sorted(glob.glob(f"{datadir}/*/"))
        self.fns = []
        for x in shard_for_rank(shards, local_rank, local_world_size):
            self.fns.extend(shard_files(x))
        self.rng = np.random.default_rng(seed)

    def __iter__(self):
        fns = list(self.fns)
        self.rng.shuffle(fns)
        for fn in fns:
            d = load_shard(fn)
            block_shuffle(d, self.rng)
            n = len(d["userid"])
            assert n % self.batch_size == 0
            for i in range(0, n, self.batch_size):
                yield {k: v[i:i + self.batch_size] for k, v in d.items()}


class FinetuneDataset:
    """FinetuneDataset (train.py:101-160): every shard file holds one user per row (arrays of shape (N, S)); only rows
    whose target weights for the finetuned medium (watch or rating) are positive are used.  Training (`shuffle`): a
    quarter of those rows per pass (partition p of 4, advancing every pass), shuffled, padded to whole batches with
    random repeats; evaluation: all rows in order, the last batch may be short."""

    def __init__(self, datadir, local_rank, local_world_size, batch_size, shuffle, finetune_medium, seed=0):
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.medium = finetune_medium
        shards = sorted(glob.glob(f"{datadir}/*/"))
        self.fns = []
        for x in shard_for_rank(shards, local_rank, local_world_size):
            self.fns.extend(shard_files(x))
        checkpoints_per_epoch = 4
        self.partition = [0, checkpoints_per_epoch]
        self.rng = np.random.default_rng(seed)

    def _eligible(self, d):
        """row indices with a positive watch or rating target weight for the finetuned medium, ascending."""
        w = d[f"{self.medium}.watch.weight"].sum(axis=1) > 0
        r = d[f"{self.medium}.rating.weight"].sum(axis=1) > 0
        return np.flatnonzero(w | r)

    def __iter__(self):
        part, nparts = self.partition
        for fn in self.fns:
            d = load_shard(fn)
            rows = self._eligible(d)
            if self.shuffle:
                rows = rows[part::nparts].copy()              # this pass's quarter of the users
                self.rng.shuffle(rows)
                short = (-len(rows)) % self.batch_size
                if short and len(rows):                        # whole batches: repeat random rows of the same quarter
                    rows = np.concatenate([rows, self.rng.choice(rows, short)])
            for i in range(0, len(rows), self.batch_size):
                take = rows[i:i + self.batch_size]
                yield {k: v[take, :] for k, v in d.items()}
        self.partition[0] = (part + 1) % nparts


def write_shards(datadir, streams, num_shards, fmt="npz"):
    """Writes one file per (shard, part) and num_tokens.txt (transformer.jl:228-239); fmt "h5" = blosc-3 HDF5."""
    total = 0
    for i, parts in enumerate(streams):
        dest = os.path.join(datadir, str(i % num_shards + 1))
        os.makedirs(dest, exist_ok=True)
        for p, d in enumerate(parts):
            if fmt == "h5":
                from . import h5
                h5.write_h5(os.path.join(dest, f"{p + 1}.h5"), d, blosc=3)
            else:
                np.savez(os.path.join(dest, f"{p + 1}.npz"), **d)
            total += len(d["userid"])
    with open(os.path.join(datadir, "num_tokens.txt"), "w") as f:
        f.write(str(total))
    return total
