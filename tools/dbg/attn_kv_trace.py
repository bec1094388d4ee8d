"""Where a dK/dV workgroup's time goes (attn_bwd_kv_dma_kernel built with -DATTN_KV_TRACE: tools/trace_attn_kv.sh): wall-clock stamps
(10 ns) of wave 0 at entry, after the first item is staged, after the item loop, at exit, and the item loop's phases summed per workgroup.
Each stamp is an s_memrealtime that drains the scalar queue, so the short phases are somewhat overstated."""
import ctypes as C, os, sys
import numpy as np
sys.argv = [sys.argv[0], "3"]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_attn as ba   # runs the launches (RSYS_LIB_PATH = the traced build)
lib = ba.lib
tr = np.zeros(16 * 8192, np.uint64)
rc = lib.rsys_attn_trace_read(tr.ctypes.data_as(C.c_void_p), C.c_ulonglong(tr.nbytes)); assert rc == 0, rc
tr = tr.reshape(8192, 16).astype(np.float64)
tr = tr[tr[:, 0] > 0]     # the workgroups of the last launch (4096 of 64 keys, or 2048 of 128 keys: attn_bwd_kv32_kernel)
slots = 1024 if len(tr) > 2048 else 512
t0, t1, t2, t3 = tr[:, 0], tr[:, 1], tr[:, 2], tr[:, 3]
items, comp = tr[:, 8], tr[:, 9]
us = lambda a: a * 0.01
print("workgroups", len(tr), " kernel span %.1f us;  sum of workgroup times / resident slots: %.1f us" % (us(t3.max() - t0.min()), us((t3 - t0).sum()) / slots))
print("items per workgroup: mean %.2f (p10 %d, p50 %d, p90 %d, max %d); computed by wave 0: mean %.2f" % (items.mean(), *np.percentile(items, [10, 50, 90]).astype(int), items.max(), comp.mean()))
for name, a in (("entry -> first item staged", t1 - t0), ("item loop", t2 - t1), ("epilogue", t3 - t2), ("whole workgroup", t3 - t0)):
    print("%-32s mean %6.2f us   p10 %6.2f  p50 %6.2f  p90 %6.2f" % (name, us(a).mean(), *us(np.percentile(a, [10, 50, 90]))))
n = items.sum()
for i, name in ((4, "issue next item's DMA + scalar loads"), (5, "arithmetic of the item"), (6, "publish (waits for the DMA)"), (7, "barrier")):
    print("%-40s %.3f us per item" % (name, us(tr[:, i].sum()) / n))
