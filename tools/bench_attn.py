"""Attention forward + backward alone at the benchmark's shape (cfg-3: 64 rows x 1024 tokens, 8 / 4 heads of 64) on the
bench's synthetic users (the tile maps' sparsity is part of the cost), through rsys_op_attention.  Run under
`rocprofv3 --kernel-trace --stats` and read the kernels' average durations (tools/ab_attn.sh alternates the values of an
environment variable, e.g. RSYS_LIB_PATH between two builds, on one box)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendersystem_amd import _lib, workload
if os.environ.get("RSYS_LIB_PATH"):            # an experimental build of the library (same-box A/B of compile-time variants)
    _lib.LIB_PATH = os.environ["RSYS_LIB_PATH"]
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
O = dev(np.zeros((B * T, H * hd), np.uint16)); dq = dev(np.zeros((B * T, Nq), np.uint16)); lse = dev(np.zeros((B, H, T), np.float32))
d_uid = dev(uid); d_tm = dev(tm)
half = hd // 2
f = 1.0 / (500000.0 ** (np.arange(0, hd, 2, dtype=np.float32) / hd)); ang = np.outer(np.arange(T, dtype=np.float32), f)
cos = dev(np.cos(ang).astype(np.float32)); sin = dev(np.sin(ang).astype(np.float32))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    _lib.check(lib.rsys_op_attention(1, B, T, H, KV, hd, qkv, d_uid, d_tm, O, lse, dO, dq, cos, sin))
print("ok")
