"""Data-parallel plumbing: one process per GPU (train.py:582-587), gradient all-reduce over RCCL/xGMI.

The data path (gradient buckets) goes through the library's own RCCL communicator
(rsys_comm_*, include/rsys.h).  The control plane -- exchanging the 128-byte RCCL id
(the reference leaves that to torchrun's store, train.py:582), barriers, the scalar
max/sum of the benchmark and the agreement on failures -- is a small TCP star around
rank 0 (`HostGroup`, plain sockets): a rank never imports torch, so the HIP runtime and
RCCL it binds are the ones librsys_hip.so was built against (/opt/rocm).  The launcher
only has to provide the usual environment: RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR,
MASTER_PORT (torchrun, or bench.py's / cli.py's own spawner).
"""
import ctypes as C
import json
import os
import socket
import struct
import time

import numpy as np

from ._lib import check, lib


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


class RendezvousError(RuntimeError):
    pass


MAX_HEADER = 1 << 20          # a frame's JSON header (a list of floats at most)
HANDSHAKE_HEADER = 4096       # the first frame of a connection: {"token", "rank"}, nothing else and no payload


def _send_frame(sock, header, raw=b""):
    """frame = !IQ (header bytes, payload bytes: 64-bit, a gathered table shard may exceed 4 GiB) + JSON header + payload"""
    head = json.dumps(dict(header, raw=len(raw))).encode()
    sock.sendall(struct.pack("!IQ", len(head), len(raw)) + head)
    if raw:
        sock.sendall(raw)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise RendezvousError("peer closed the connection (another rank exited)")
        buf += chunk
    return bytes(buf)


def _recv_frame(sock, max_header=MAX_HEADER, max_raw=None):
    """the lengths are checked BEFORE anything is allocated (`max_raw` None: trusted peer that passed the handshake)"""
    nh, nr = struct.unpack("!IQ", _recv_exact(sock, 12))
    if nh > max_header or (max_raw is not None and nr > max_raw):
        raise RendezvousError(f"frame of {nh} + {nr} bytes exceeds the limit of this stage of the protocol")
    header = json.loads(_recv_exact(sock, nh).decode())
    return header, (_recv_exact(sock, nr) if nr else b"")


def _is_loopback(addr):
    """True for the loopback literals a single-node launcher passes as MASTER_ADDR (127.0.0.0/8, ::1, localhost)"""
    a = (addr or "").strip().lower()
    return a == "localhost" or a == "::1" or a.startswith("127.")


def rendezvous_ports():
    """Candidate ports of rank 0's listener.  RSYS_RDZV_PORT names one (bench.py's own spawner sets it); under torchrun
    MASTER_PORT itself is held by the agent's store, so the candidates are the 16 ports after it and the handshake
    (session token + world size) tells the right listener from anything else that may sit on one of them."""
    if os.environ.get("RSYS_RDZV_PORT"):
        return [int(os.environ["RSYS_RDZV_PORT"])]
    base = int(os.environ.get("MASTER_PORT", "29500"))
    return [base + 1 + k for k in range(16)]


class HostGroup:
    """CPU-side process group over TCP: rank 0 listens, every other rank holds one connection to it; each collective is
    one request per rank and one reply (star).  Messages carry an operation counter, so ranks that fall out of step
    fail with an error instead of exchanging the wrong values; a rank that dies closes its socket, which every other
    rank sees as an error in its next collective (nobody is left waiting).  world == 1 opens nothing."""

    def __init__(self, rank=None, world=None, timeout=None):
        r, w, _ = env_rank()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.timeout = float(os.environ.get("RSYS_RDZV_TIMEOUT", "600")) if timeout is None else timeout
        self.seq = 0
        self.peers = []      # rank 0: sockets of ranks 1 .. world-1, by rank
        self.sock = None     # other ranks: the connection to rank 0
        self.listener = None
        if self.world > 1:
            addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
            token = f"{os.environ.get('TORCHELASTIC_RUN_ID', '')}:{os.environ.get('MASTER_PORT', '')}:{self.world}"
            (self._serve if self.rank == 0 else self._join)(addr, token)

    # ---- connection set-up
    def _serve(self, addr, token):
        err = None
        for port in rendezvous_ports():
            ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                # rank 0 listens on loopback only when the rendezvous address IS loopback (127.0.0.1 / localhost under
                # launch_local / torchrun --standalone); for any other MASTER_ADDR -- e.g. a host name that /etc/hosts maps to
                # 127.0.1.1 on rank 0 but to the real address elsewhere -- on every interface (the handshake token rejects strangers)
                ls.bind((addr if _is_loopback(addr) else "", port))
            except OSError as e:
                err = e
                ls.close()
                continue
            ls.listen(self.world)
            self.listener = ls
            break
        if self.listener is None:
            raise RendezvousError(f"rank 0 cannot bind any rendezvous port {rendezvous_ports()}: {err}")
        peers = {}
        deadline = time.monotonic() + self.timeout
        while len(peers) < self.world - 1:
            self.listener.settimeout(max(0.1, deadline - time.monotonic()))
            try:
                conn, _ = self.listener.accept()
            except socket.timeout:
                raise RendezvousError(f"rendezvous timed out: {len(peers) + 1} of {self.world} ranks arrived") from None
            conn.settimeout(2.0)          # (a stranger on the port costs the rendezvous two seconds, not ten)
            try:
                hello, _ = _recv_frame(conn, max_header=HANDSHAKE_HEADER, max_raw=0)
                ok = hello.get("token") == token and 0 < int(hello.get("rank", -1)) < self.world and hello["rank"] not in peers
                _send_frame(conn, {"ok": bool(ok)})
            except (OSError, ValueError, RendezvousError, struct.error):
                ok = False
            if not ok:
                conn.close()
                continue
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            conn.settimeout(self.timeout)
            peers[int(hello["rank"])] = conn
        self.peers = [peers[r] for r in range(1, self.world)]

    def _join(self, addr, token):
        deadline = time.monotonic() + self.timeout
        while True:
            for port in rendezvous_ports():
                try:
                    s = socket.create_connection((addr, port), timeout=5.0)
                except OSError:
                    continue
                try:
                    s.settimeout(10.0)
                    _send_frame(s, {"token": token, "rank": self.rank})
                    reply, _ = _recv_frame(s, max_header=HANDSHAKE_HEADER, max_raw=0)
                    if reply.get("ok"):
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        s.settimeout(self.timeout)
                        self.sock = s
                        return
                except (OSError, ValueError, RendezvousError, struct.error):
                    pass
                s.close()
            if time.monotonic() > deadline:
                raise RendezvousError(f"rank {self.rank}: no rendezvous listener of rank 0 on {addr}:{rendezvous_ports()}")
            time.sleep(0.2)

    # ---- one collective = gather at rank 0, combine, scatter the result
    def _collective(self, op, header, raw, combine):
        self.seq += 1
        header = dict(header, op=op, seq=self.seq)
        try:
            if self.rank != 0:
                _send_frame(self.sock, header, raw)
                out, out_raw = _recv_frame(self.sock)
                if "error" in out:
                    raise RendezvousError(out["error"])
                return out, out_raw
            parts = [(header, raw)]
            for r, conn in enumerate(self.peers, start=1):
                h, b = _recv_frame(conn)
                if h.get("op") != op or h.get("seq") != self.seq:
                    msg = f"ranks out of step: rank 0 is in {op}#{self.seq}, rank {r} in {h.get('op')}#{h.get('seq')}"
                    for c in self.peers:
                        _send_frame(c, {"error": msg})
                    raise RendezvousError(msg)
                parts.append((h, b))
            out, out_raw = combine(parts)
            for conn in self.peers:
                _send_frame(conn, out, out_raw)
            return out, out_raw
        except (OSError, struct.error) as e:
            raise RendezvousError(f"rank {self.rank}: {op} failed: {e}") from None

    def barrier(self):
        if self.world > 1:
            self._collective("barrier", {}, b"", lambda parts: ({}, b""))

    def broadcast_bytes(self, data, src=0):
        if self.world == 1:
            return data
        _, raw = self._collective("bcast", {"src": src}, bytes(data) if self.rank == src else b"",
                                  lambda parts: ({}, parts[src][1]))
        return raw

    def all_reduce(self, values, op="sum"):
        """element-wise sum / max / min of a short list of floats over the ranks (float64, summed in rank order)"""
        vals = [float(v) for v in values]
        if self.world == 1:
            return vals
        red = {"sum": np.sum, "max": np.max, "min": np.min}[op]

        def combine(parts):
            m = np.array([h["v"] for h, _ in parts], np.float64)
            return {"v": [float(x) for x in red(m, axis=0)]}, b""
        out, _ = self._collective("reduce_" + op, {"v": vals}, b"", combine)
        return out["v"]

    def all_gather_bytes(self, data):
        """every rank's byte string, in rank order, on every rank (checkpoints of a row-sharded table: host plumbing)"""
        data = bytes(data)
        if self.world == 1:
            return [data]

        def combine(parts):
            return {"sizes": [len(b) for _, b in parts]}, b"".join(b for _, b in parts)
        out, raw = self._collective("all_gather", {}, data, combine)
        chunks, at = [], 0
        for n in out["sizes"]:
            chunks.append(raw[at:at + n]); at += n
        return chunks

    def all_reduce_array(self, arr):
        """in-place float32 sum over the ranks through host memory (debugging aid: HostComm)"""
        if self.world == 1:
            return arr

        def combine(parts):
            acc = np.frombuffer(parts[0][1], np.float32).copy()
            for _, b in parts[1:]:
                acc += np.frombuffer(b, np.float32)
            return {}, acc.tobytes()
        _, raw = self._collective("reduce_array", {"n": int(arr.size)}, np.ascontiguousarray(arr, np.float32).tobytes(), combine)
        arr[...] = np.frombuffer(raw, np.float32).reshape(arr.shape)
        return arr

    def close(self):
        for c in self.peers + [x for x in (self.sock, self.listener) if x is not None]:
            try:
                c.close()
            except OSError:
                pass
        self.peers, self.sock, self.listener = [], None, None


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

    def info(self):
        """{rank, ranks, transport, rccl_version}: what a SCALE record needs to say which collectives library ran (bench.py `comm`)"""
        out = (C.c_int32 * 4)()
        check(lib().rsys_comm_info(self._h, out))
        v = int(out[3])
        # ncclGetVersion's code: major * 10000 + minor * 100 + patch from 2.9 on (major * 1000 + minor * 100 + patch before)
        ver = None if not v else (f"{v // 10000}.{v // 100 % 100}.{v % 100}" if v >= 10000 else f"{v // 1000}.{v // 100 % 10}.{v % 100}")
        return {"rank": int(out[0]), "ranks": int(out[1]), "transport": {0: "none", 1: "rccl", 2: "in-process"}[int(out[2])], "rccl_version": ver}

    def grad_schedule(self, model, cap=64):
        """the last optimizer step's gradient reduction as enqueued: [(first element, one past the last, phase)] (rsys_grad_sync_schedule)"""
        out = (C.c_int64 * (3 * cap))(); n = C.c_int32()
        check(lib().rsys_grad_sync_schedule(model._h, out, cap, C.byref(n)))
        k = min(int(n.value), cap)
        return [(int(out[3 * i]), int(out[3 * i + 1]), int(out[3 * i + 2])) for i in range(k)]

    def debug_delay(self, microseconds):
        """tests: a kernel that spins this long on the communicator's stream (a collective that starts late)"""
        check(lib().rsys_comm_debug_delay(self._h, int(microseconds)))

    def close(self):
        if self._h:
            lib().rsys_comm_destroy(self._h)
            self._h = C.c_void_p()


class LocalGroup:
    """`world` ranks of THIS process on one GPU, each driven by its own host thread (rsys_local_group_create): the
    collectives are device copies between the ranks' buffers.  For tests of the multi-rank arithmetic on a one-GPU box
    (two RCCL ranks cannot share a GPU); `LocalComm(group, rank)` is a rank's communicator, same surface as `Comm`."""

    def __init__(self, world, device=0):
        self.world = world
        self._h = C.c_void_p()
        check(lib().rsys_local_group_create(world, device, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().rsys_local_group_destroy(self._h)
            self._h = C.c_void_p()


class LocalComm(Comm):
    def __init__(self, group, rank):     # noqa: super().__init__ is RCCL's rendezvous
        self.hg = None
        self.rank, self.world = rank, group.world
        self._h = C.c_void_p()
        check(lib().rsys_comm_init_local(group._h, rank, C.byref(self._h)))


class HostComm:
    """Gradient all-reduce through host memory over the TCP control plane (same surface as Comm): a debugging aid for
    machines whose RCCL cannot initialise.  Never chosen silently: `make_comm` raises unless RSYS_ALLOW_HOST_ALLREDUCE=1
    is set, and the class says what it is on stderr (it is orders of magnitude slower than RCCL over xGMI)."""

    def __init__(self, host_group, reason=""):
        import sys
        self.hg = host_group
        self.rank, self.world = host_group.rank, host_group.world
        if self.rank == 0:
            print(f"[recommendersystem_amd.dist] RCCL communicator unavailable ({reason}); "
                  "falling back to a HOST all-reduce over TCP", file=sys.stderr)

    def self_test(self):
        assert self.hg.all_reduce([1.0], "sum")[0] == float(self.world)

    def begin_grad_sync(self, model):
        pass

    def all_reduce_grads(self, model):
        ptr = C.c_void_p(); n = C.c_int64()
        check(lib().rsys_grad_buffer(model._h, C.byref(ptr), C.byref(n)))
        host = np.empty(n.value, np.float32)
        check(lib().rsys_dev_d2h(host.ctypes.data, ptr, host.nbytes))
        self.hg.all_reduce_array(host)
        check(lib().rsys_dev_h2d(ptr, host.ctypes.data, host.nbytes))

    def all_reduce_sum(self, values):
        return self.hg.all_reduce(values, "sum")

    def close(self):
        pass


def _no_rccl(host_group, reason):
    if os.environ.get("RSYS_ALLOW_HOST_ALLREDUCE") == "1":
        return HostComm(host_group, reason)
    raise RuntimeError(f"RCCL communicator unavailable on at least one rank ({reason}); the gradient all-reduce has no "
                       "silent host path (set RSYS_ALLOW_HOST_ALLREDUCE=1 to debug with an all-reduce through host memory)")


def make_comm(host_group, device):
    """RCCL communicator of this rank; every rank takes the same branch (the outcome of the RCCL attempt is agreed on
    with a MIN reduction over the control plane, so a failure raises on all ranks instead of leaving some inside a collective)."""
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


class ReplicaMismatch(RuntimeError):
    pass


def assert_replicas_equal(model, comm, what=""):
    """The reference's DDP constructor broadcasts rank 0's parameters to every rank (transformer.py:678-682).  Here every rank builds
    the same parameters itself (same seed, same checkpoint file) and the ranks VERIFY it: each rank's checksum words
    (`model.param_checksum()`: fp64 sum, fp64 sum of squares and an integer sum of the bit patterns, computed in a fixed order on the
    device -- or the words themselves, for a caller that has no device model) travel in one slot per rank of a SUM all-reduce (the other
    slots are zero, so the sums are exact), every rank then holds every rank's words and checks minimum == maximum.  A mismatch raises
    `ReplicaMismatch` on EVERY rank, with the differing ranks named: nobody trains on.  Called after init, after a resume and at the end
    of every epoch (train.train).  comm = a `Comm` / `LocalComm` / `HostComm` (all_reduce_sum) or a `HostGroup`; None or a world of one: nothing to do."""
    world = 1 if comm is None else comm.world
    if world <= 1:
        return None
    words = model.param_checksum() if hasattr(model, "param_checksum") else [float(w) for w in model]
    k = len(words)
    slots = [0.0] * (k * world)
    slots[k * comm.rank:k * comm.rank + k] = words
    reduce = comm.all_reduce_sum if hasattr(comm, "all_reduce_sum") else (lambda v: comm.all_reduce(v, "sum"))
    got = np.asarray(reduce(slots), np.float64).reshape(world, k)
    if not (got.min(axis=0) == got.max(axis=0)).all():     # (NaN parameters fail this too: NaN != NaN)
        ref = got[0]
        odd = [r for r in range(world) if not (got[r] == ref).all()]
        raise ReplicaMismatch(f"replicas differ{' ' + what if what else ''}: ranks {odd} do not hold rank 0's parameters "
                              f"(checksum words by rank: {got.tolist()})")
    return got[0].tolist()


def shard_for_rank(shards, local_rank, local_world_size):
    """train.py:46-51: shard directory i goes to rank i % world; the count must divide evenly."""
    assert len(shards) % local_world_size == 0
    return [x for i, x in enumerate(shards) if i % local_world_size == local_rank]


def launch_local(nproc, argv, env=None, poll=0.2, grace=10.0):
    """`torchrun --standalone --nproc_per_node=N` (entrypoint.sh:25) for one node, without torch: start `nproc` fresh
    processes of `argv` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / RSYS_RDZV_PORT set, rank 0 on
    this process's stdout and the other ranks' stdout on stderr.  The caller must not have touched the GPU (the children
    are ordinary child processes, nothing is exec'd over a process that has).  As soon as one rank exits non-zero the
    others are terminated (no orphan is left inside a collective); returns the first non-zero exit code, else 0."""
    import subprocess
    import sys
    ls = socket.socket(); ls.bind(("127.0.0.1", 0)); port = ls.getsockname()[1]; ls.close()
    procs = []
    for r in range(nproc):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RSYS_RDZV_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(list(argv), env=e, stdout=None if r == 0 else sys.stderr))
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(poll)
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
    if rc != 0:
        for p in live:
            p.terminate()
        deadline = time.monotonic() + grace
        for p in live:
            try:
                p.wait(timeout=max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc
