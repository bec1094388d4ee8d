"""CPU, world_size 2: the N>1 host path (TCP rendezvous from the launcher's env, id broadcast, scalar reductions,
DDP-mean semantics of the gradient all-reduce, reduce_mean, shard assignment) with torch un-importable in the ranks;
the spawner of `bench.py --gpus N` and its failure behaviour."""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_world2_host_group_without_torch(tmp_path):
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
        assert r["bcast_ok"] and r["array_sum_ok"] and r["torch_blocked"] and r["min"] == [0.0]
        assert r["gathered"] == [3, 5, True]
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


def test_world2_under_torchrun_style_env_finds_a_port_next_to_the_store(tmp_path):
    """Under torchrun MASTER_PORT itself belongs to the agent's store: without RSYS_RDZV_PORT the ranks meet on one of the
    ports after it, also when the first candidate is taken by something that does not speak the handshake."""
    port = _free_port()
    squat = socket.socket(); squat.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    try:
        squat.bind(("127.0.0.1", port + 1)); squat.listen(4)          # a stranger on the first candidate
    except OSError:
        squat = None
    out = str(tmp_path / "res")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", TORCHELASTIC_RUN_ID="t1")
        env.pop("RSYS_RDZV_PORT", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), out], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    if squat is not None:
        squat.close()
    assert all(json.load(open(out + f".{r}.json"))["bcast_ok"] for r in range(2))


def test_collective_fails_on_every_rank_when_a_peer_dies():
    """A rank that exits closes its socket; the others get an error from their next collective instead of waiting."""
    import textwrap
    port = _free_port()
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        from recommendersystem_amd import dist
        hg = dist.HostGroup()
        hg.barrier()
        if hg.rank == 1:
            os._exit(7)
        try:
            hg.barrier()
        except dist.RendezvousError:
            sys.exit(5)
        sys.exit(0)
    """ % ROOT)
    procs = []
    for rank in range(3):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RSYS_RDZV_PORT=str(port), RSYS_RDZV_TIMEOUT="60")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env))
    assert [p.wait(timeout=120) for p in procs] == [5, 7, 5]


def test_replica_consistency_check_fails_on_every_rank_when_one_rank_differs(tmp_path):
    """SURVEY 2.4 C1 / transformer.py:678-682: DDP broadcasts rank 0's parameters; this package initialises every rank from the same
    seed and CHECKS (dist.assert_replicas_equal: one slot per rank of a SUM all-reduce, min == max on every rank).  World of three over
    the TCP control plane, checksum words of a host buffer standing in for the device model's (`model.param_checksum`, GPU test in
    tests/test_gpu_shard.py): equal buffers pass and every rank returns the same words; with ONE element of rank 1's buffer off by one
    ulp every rank -- not only rank 1 -- raises ReplicaMismatch naming rank 1."""
    import textwrap
    port = _free_port()
    code = textwrap.dedent("""
        import json, os, sys
        import numpy as np
        sys.modules["torch"] = None
        sys.path.insert(0, %r)
        from recommendersystem_amd import dist
        hg = dist.HostGroup()
        P = np.random.default_rng(7).standard_normal(100003).astype(np.float32)
        def words(p):
            bits = int((p.view(np.uint32).astype(np.uint64) * (np.arange(p.size, dtype=np.uint64) %% 1024 + 1)).sum(dtype=np.uint64))
            return [float(p.astype(np.float64).sum()), float((p.astype(np.float64) ** 2).sum()), float(bits & 0xffffffff), float(bits >> 32)]
        res = {"ok": dist.assert_replicas_equal(words(P), hg, "after init")}
        if hg.rank == 1:
            P[54321] = np.nextafter(P[54321], np.float32(10.0))
        try:
            dist.assert_replicas_equal(words(P), hg, "after epoch 0")
            res["second"] = "passed"
        except dist.ReplicaMismatch as e:
            res["second"] = str(e)
        res["single"] = dist.assert_replicas_equal(words(P), None)
        json.dump(res, open(%r + ".%%d" %% hg.rank, "w"))
        hg.close()
    """ % (ROOT, str(tmp_path / "rc")))
    procs = []
    for rank in range(3):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RSYS_RDZV_PORT=str(port), RSYS_RDZV_TIMEOUT="60")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env))
    assert [p.wait(timeout=120) for p in procs] == [0, 0, 0]
    res = [json.load(open(str(tmp_path / "rc") + f".{r}")) for r in range(3)]
    assert res[0]["ok"] == res[1]["ok"] == res[2]["ok"] and len(res[0]["ok"]) == 4
    for r in res:
        assert r["second"].startswith("replicas differ after epoch 0: ranks [1] do not hold rank 0's parameters"), r["second"]
        assert r["single"] is None


def test_launch_local_relays_and_stops_everyone_on_failure(tmp_path):
    from recommendersystem_amd.dist import launch_local
    ok = tmp_path / "ok.py"
    ok.write_text("import os, sys\nsys.path.insert(0, %r)\nfrom recommendersystem_amd import dist\nhg = dist.HostGroup()\n"
                  "s = hg.all_reduce([hg.rank + 1.0])[0]\nopen(os.path.join(%r, 'r%%d' %% hg.rank), 'w').write(str(s))\nhg.close()\n"
                  % (ROOT, str(tmp_path)))
    assert launch_local(3, [sys.executable, str(ok)]) == 0
    assert [open(tmp_path / f"r{r}").read() for r in range(3)] == ["6.0"] * 3
    bad = tmp_path / "bad.py"
    bad.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(9)\ntime.sleep(600)\n")
    import time
    t0 = time.time()
    assert launch_local(3, [sys.executable, str(bad)], grace=5.0) == 9
    assert time.time() - t0 < 60                     # the sleeping ranks were terminated, not waited for


def test_bench_gpus2_without_gpus_fails_cleanly_on_every_rank():
    """`python bench.py --gpus 2` with no torchrun around it spawns its own ranks; on a machine with fewer GPUs than ranks
    every rank agrees on the failure over the control plane and exits non-zero -- no rank is left in a collective."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "RSYS_RDZV_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""                  # also on a GPU box: no device for anybody
    env["ROCR_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    assert p.stdout.strip() == ""
    assert p.stderr.count("ranks need one each") == 2


def test_rank0_binds_loopback_only_for_a_loopback_master_addr(monkeypatch):
    """with a host name as MASTER_ADDR (which rank 0's /etc/hosts may map to 127.0.1.1 while the other nodes resolve the
    real address) rank 0 listens on every interface; only a loopback literal restricts the listener."""
    import threading

    from recommendersystem_amd import dist

    assert dist._is_loopback("127.0.0.1") and dist._is_loopback("localhost") and dist._is_loopback("127.0.1.1")
    assert not dist._is_loopback(socket.gethostname() or "node0") or socket.gethostname() == "localhost"
    assert not dist._is_loopback("10.0.0.5") and not dist._is_loopback("")
    port = _free_port()
    for k, v in dict(MASTER_ADDR="some-node-name.invalid", MASTER_PORT=str(port), RSYS_RDZV_PORT=str(port), TORCHELASTIC_RUN_ID="bind").items():
        monkeypatch.setenv(k, v)
    groups = [None]; err = []

    def rank0():
        try:
            groups[0] = dist.HostGroup(0, 2, timeout=20)
        except BaseException as e:   # noqa: BLE001
            err.append(e)
    t0 = threading.Thread(target=rank0); t0.start()
    deadline = time.time() + 10
    while True:                                       # rank 1 reaches the listener through loopback: it is bound to every interface
        try:
            s = socket.create_connection(("127.0.0.1", port), timeout=1.0); break
        except OSError:
            assert time.time() < deadline and not err, err; time.sleep(0.05)
    dist._send_frame(s, {"token": f"bind:{port}:2", "rank": 1})
    assert dist._recv_frame(s)[0]["ok"] is True
    t0.join(20)
    assert not err, err
    assert groups[0].listener.getsockname()[0] == "0.0.0.0"
    s.close(); groups[0].listener.close()
    for c in groups[0].peers:
        c.close()


def test_rendezvous_rejects_oversized_and_foreign_handshakes_without_allocating(monkeypatch):
    """rank 0 honoured two 32-bit length prefixes (up to 4 GiB each) before the token check, on every interface.  Now the
    first frame of a connection is capped (4 KB header, no payload) before anything is read, the listener binds the rendezvous
    address, and payload lengths are 64-bit.  A stranger that announces a 3 GiB frame, one that sends garbage and one with the
    wrong token are dropped; the real rank 1 still joins, and a > 4 GiB-capable length field round-trips."""
    import struct
    import threading

    from recommendersystem_amd import dist

    port = _free_port()
    for k, v in dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RSYS_RDZV_PORT=str(port), TORCHELASTIC_RUN_ID="hs").items():
        monkeypatch.setenv(k, v)
    groups = [None, None]; err = []

    def rank(r):
        try:
            groups[r] = dist.HostGroup(r, 2, timeout=30)
        except BaseException as e:   # noqa: BLE001
            err.append(e)
    t0 = threading.Thread(target=rank, args=(0,)); t0.start()
    deadline = time.time() + 10
    while True:                                       # wait for the listener, then three strangers before the real rank
        try:
            s = socket.create_connection(("127.0.0.1", port), timeout=1.0); break
        except OSError:
            assert time.time() < deadline; time.sleep(0.05)
    s.sendall(struct.pack("!IQ", 3 << 30, 3 << 30)); time.sleep(0.1)
    try:
        assert s.recv(16) == b""                     # closed without a reply: nothing of the announced 6 GiB was awaited
    except OSError:
        pass
    s.close()
    s = socket.create_connection(("127.0.0.1", port), timeout=1.0); s.sendall(b"GET / HTTP/1.0\r\n\r\n"); s.close()
    s = socket.create_connection(("127.0.0.1", port), timeout=1.0)
    dist._send_frame(s, {"token": "someone-else", "rank": 1})
    assert dist._recv_frame(s)[0] == {"ok": False, "raw": 0}
    s.close()
    t1 = threading.Thread(target=rank, args=(1,)); t1.start()
    t0.join(40); t1.join(40)
    assert not err, err
    assert groups[0].listener.getsockname()[0] == "127.0.0.1"          # not 0.0.0.0
    out = [None, None]
    th = [threading.Thread(target=lambda r=r: out.__setitem__(r, groups[r].all_reduce([r + 1.0], "sum"))) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(20)
    assert out == [[3.0], [3.0]]
    assert struct.calcsize("!IQ") == 12 and struct.unpack("!IQ", struct.pack("!IQ", 7, 5 << 32))[1] == 5 << 32
    for g in groups:
        g.close()
