"""Forward / backward attention outputs of two builds on the same operands: python tools/dbg/attn_cmp.py <lib_a.so> <lib_b.so> [B T H KV hd dtype]"""
import os, subprocess, sys, tempfile
import numpy as np
if len(sys.argv) >= 3 and sys.argv[1] != "--child":
    shape = sys.argv[3:] or ["2", "128", "2", "1", "64", "0"]
    outs = []
    for lib in sys.argv[1:3]:
        f = tempfile.mktemp(suffix=".npz")
        subprocess.check_call([sys.executable, __file__, "--child", lib, f] + shape)
        outs.append(np.load(f))
    for k in outs[0].files:
        a, b = outs[0][k].astype(np.float64), outs[1][k].astype(np.float64)
        bad = np.argwhere(np.abs(a - b) > 1e-3 * max(np.abs(a).max(), 1e-9))
        print(k, "max diff", np.abs(a - b).max(), "of", np.abs(a).max(), "mismatching elements", len(bad), "rows", sorted(set(bad[:, 0].tolist()))[:40] if len(bad) else "")
    sys.exit(0)
import ctypes as C
lib_path, out = sys.argv[2], sys.argv[3]
B, T, H, KV, hd, dtype = [int(x) for x in sys.argv[4:10]]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from recommendersystem_amd import _lib
_lib.LIB_PATH = lib_path
lib = _lib.lib()
rng = np.random.default_rng(1)
Nq = (H + 2 * KV) * hd
uid = np.sort(rng.integers(1, 4, (B, T)), axis=1).astype(np.int32).reshape(-1)
tm = (rng.integers(1, 3, B * T) * (rng.random(B * T) < 0.2)).astype(np.int32)
def dev(a):
    p = C.c_void_p(); assert lib.rsys_dev_alloc(C.byref(p), a.nbytes) == 0; lib.rsys_dev_h2d(p, a.ctypes.data, a.nbytes); return p
def _round_bf16(a):   # round to nearest even, as the device does
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    return ((u + (((u >> 16) & 1) + 0x7FFF)) & 0xFFFF0000).view(np.float32)
# (RSYS_CMP_ROUND=1: the fp32 run takes the bf16-rounded operands of the bf16 runs -- the exact result of the same inputs)
cvt = (lambda a: (_round_bf16(a).view(np.uint32) >> 16).astype(np.uint16)) if dtype == 1 else ((lambda a: _round_bf16(a)) if os.environ.get("RSYS_CMP_ROUND") == "1" else (lambda a: np.ascontiguousarray(a, np.float32)))
back = (lambda u: (u.astype(np.uint32) << 16).view(np.float32)) if dtype == 1 else (lambda u: u)
et = np.uint16 if dtype == 1 else np.float32
qkv = dev(cvt(rng.standard_normal((B * T, Nq)))); dO = dev(cvt(rng.standard_normal((B * T, H * hd))))
O = dev(np.zeros((B * T, H * hd), et)); dq = dev(np.zeros((B * T, Nq), et)); lse = dev(np.zeros((B, H, T), np.float32))
f = 1.0 / (500000.0 ** (np.arange(0, hd, 2, dtype=np.float32) / hd)); ang = np.outer(np.arange(T, dtype=np.float32), f)
cos = dev(np.cos(ang).astype(np.float32)); sin = dev(np.sin(ang).astype(np.float32))
_lib.check(lib.rsys_op_attention(dtype, B, T, H, KV, hd, qkv, dev(uid), dev(tm), O, lse, dO, dq, cos, sin))
o = np.empty((B * T, H * hd), et); g = np.empty((B * T, Nq), et); l_ = np.empty((B, H, T), np.float32)
lib.rsys_dev_d2h(o.ctypes.data, O, o.nbytes); lib.rsys_dev_d2h(g.ctypes.data, dq, g.nbytes); lib.rsys_dev_d2h(l_.ctypes.data, lse, l_.nbytes)
np.savez(out, O=back(o), dqkv=back(g), lse=l_.reshape(B * H, T))
