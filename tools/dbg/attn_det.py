import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from recommendersystem_amd import _lib, workload
if os.environ.get('RSYS_LIB_PATH'): _lib.LIB_PATH = os.environ['RSYS_LIB_PATH']
lib = _lib.lib()
cfg = workload.make_config("cfg3")
B, S, H, KV, hd = 64, cfg["max_sequence_length"], cfg["num_heads"], cfg["num_kv_heads"], cfg["embed_dim"] // cfg["num_heads"]
T = 2 * S
d = workload.make_batch(cfg, B, 0xD47A, mu=4.6, sigma=1.0)
uid = np.repeat(np.asarray(d["userid"], np.int32).reshape(-1), 2)
rng = np.random.default_rng(0)
tm = np.repeat((np.asarray(d["token_mask_ids"]).reshape(-1) * (rng.random(B * S) < 0.1)).astype(np.int32), 2)
Nq = (H + 2 * KV) * hd
def dev(a):
    p = C.c_void_p(); assert lib.rsys_dev_alloc(C.byref(p), a.nbytes) == 0; lib.rsys_dev_h2d(p, a.ctypes.data, a.nbytes); return p
bf = lambda a: (np.ascontiguousarray(a, np.float32).view(np.uint32) >> 16).astype(np.uint16)
qkv = dev(bf(rng.standard_normal((B * T, Nq)).astype(np.float32))); dO = dev(bf(rng.standard_normal((B * T, H * hd)).astype(np.float32)))
d_uid = dev(uid); d_tm = dev(tm)
f = 1.0 / (500000.0 ** (np.arange(0, hd, 2, dtype=np.float32) / hd)); ang = np.outer(np.arange(T, dtype=np.float32), f)
cos = dev(np.cos(ang).astype(np.float32)); sin = dev(np.sin(ang).astype(np.float32))
outs = []
NREP = int(os.environ.get('NREP', '8'))
for rep in range(NREP):
    O = dev(np.full((B * T, H * hd), 0x7fc0, np.uint16)); dq = dev(np.full((B * T, Nq), 0x7fc0, np.uint16)); lse = dev(np.zeros((B, H, T), np.float32))
    _lib.check(lib.rsys_op_attention(1, B, T, H, KV, hd, qkv, d_uid, d_tm, O, lse, dO, dq, cos, sin))
    o = np.empty((B * T, H * hd), np.uint16); g = np.empty((B * T, Nq), np.uint16); l = np.empty((B, H, T), np.float32)
    lib.rsys_dev_d2h(o.ctypes.data, O, o.nbytes); lib.rsys_dev_d2h(g.ctypes.data, dq, g.nbytes); lib.rsys_dev_d2h(l.ctypes.data, lse, l.nbytes)
    outs.append((o, g, l))
    for p in (O, dq, lse): lib.rsys_dev_free(p)
import collections
f32 = lambda u: (u.astype(np.uint32) << 16).view(np.float32)
stack = np.stack([o[1] for o in outs])           # [rep][row][col]
ref = outs[0][1] if NREP < 3 else np.where(stack[0] == stack[1], stack[0], stack[2])   # majority of the first three
nbad = 0
for rep in range(NREP):
    idx = np.argwhere(stack[rep] != ref)
    if len(idx) == 0: continue
    nbad += 1
    rows = sorted(set(idx[:, 0].tolist())); cols = sorted(set(idx[:, 1].tolist()))
    print("rep", rep, "dqkv differs in", len(idx), "elements rows", rows[0], "..", rows[-1], "(", len(rows), ") cols", cols, "-> head", cols[0] // hd, "d", cols[0] % hd, "row in q tile", rows[0] % 64, "q tile", (rows[0] % T) // 64, "user row", rows[0] // T)
    if nbad <= 6:
        c = cols[0]
        for r_ in rows[:16]:
            print("    row", r_, "good", [float(x) for x in f32(ref[r_, c - 2:c + 2])], "bad", [float(x) for x in f32(stack[rep][r_, c - 2:c + 2])], "uid", int(uid[r_]), "tm", int(tm[r_]))
for rep in range(1, NREP):
    for name, k in (("O", 0), ("lse", 2)):
        if (outs[0][k] != outs[rep][k]).any(): print("rep", rep, name, "differs")
print("glitched reps:", nbad, "of", NREP)
print("nan in O:", int((outs[0][0] == 0x7fc0).sum()), "nan in dqkv:", int((outs[0][1] == 0x7fc0).sum()))
print("done")
