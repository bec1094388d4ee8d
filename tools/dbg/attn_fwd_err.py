"""How far the two bf16 forward kernels are from the fp32 kernel on the same (bf16-rounded) operands:
python tools/dbg/attn_fwd_err.py [B T H KV hd]   -> max / rms error of O, lse and dq / dk / dv for RSYS_ATTN_FWD32 = 1 and 0"""
import os, subprocess, sys, tempfile
import numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
lib = os.path.join(root, "recommendersystem_amd", "librsys_hip.so")
shape = sys.argv[1:] or ["8", "1024", "8", "4", "64"]
def run(dtype, env):
    f = tempfile.mktemp(suffix=".npz")
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "dbg", "attn_cmp.py"), "--child", lib, f] + shape + [dtype], env=dict(os.environ, **env))
    return np.load(f)
ref = run("0", {"RSYS_CMP_ROUND": "1"})
H, KV, hd = int(shape[2]), int(shape[3]), int(shape[4])
for name, env in (("fwd32 (default)", {"RSYS_ATTN_FWD32": "1"}), ("64-query kernel", {"RSYS_ATTN_FWD32": "0"})):
    z = run("1", env)
    out = []
    for k, sl in (("O", None), ("lse", None), ("dq", slice(0, H * hd)), ("dk", slice(H * hd, (H + KV) * hd)), ("dv", slice((H + KV) * hd, None))):
        a = (z["dqkv"][:, sl] if sl is not None else z[k]).astype(np.float64); b = (ref["dqkv"][:, sl] if sl is not None else ref[k]).astype(np.float64)
        out.append(f"{k}: max {np.abs(a - b).max() / np.abs(b).max():.3e} rms {np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()):.3e}")
    print(name, " | ".join(out))
