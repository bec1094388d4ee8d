"""Phases of the dK/dV item loop (wave 0 of every workgroup; traced build of attention.hip): issue of the next-next item's global loads,
the item's arithmetic, the store of the next item into LDS (waits for its loads), the barrier."""
import ctypes as C, os, sys
import numpy as np
sys.argv = [sys.argv[0], "3"]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_attn as ba
lib = ba.lib
tr = np.zeros(8 * 8192, np.uint64)
rc = lib.rsys_attn_trace_read(tr.ctypes.data_as(C.c_void_p), C.c_ulonglong(tr.nbytes)); assert rc == 0, rc
tr = tr.reshape(8192, 8)[:4096].astype(np.float64)
items = tr[:, 4].sum(); comp = tr[:, 5].sum()
print("items", int(items), "computed by wave 0", int(comp))
for i, name in enumerate(("issue loads of item +2", "arithmetic of the item", "store item +1 to LDS (waits for its loads)", "barrier")):
    print("%-46s %.3f us per item" % (name, tr[:, i].sum() * 0.01 / items))
print("sum %.3f us per item" % (tr[:, :4].sum() * 0.01 / items))
