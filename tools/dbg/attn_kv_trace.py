"""Where a dK/dV workgroup's time goes (build of attention.hip with -DATTN_KV_TRACE, RSYS_LIB_PATH=that library): wall-clock stamps
(10 ns) at kernel entry, after the K / V fragments and tile maps arrived, after the first staged item, after the item loop, at exit."""
import ctypes as C, os, sys
import numpy as np
sys.argv = [sys.argv[0], "3"]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_attn as ba   # runs the launches
lib = ba.lib
tr = np.zeros(8 * 8192, np.uint64)
rc = lib.rsys_attn_trace_read(tr.ctypes.data_as(C.c_void_p), C.c_ulonglong(tr.nbytes)); assert rc == 0, rc
tr = tr.reshape(8192, 8)[:4096].astype(np.int64)
t0, t1, t2, t3, t4, items, comp = [tr[:, i] for i in range(7)]
us = lambda a: a * 0.01
print("workgroups", len(tr), "kernel span %.1f us" % us(t4.max() - t0.min()))
print("items per workgroup: mean %.2f  (p10 %d, p50 %d, p90 %d, max %d); computed by wave 0: mean %.2f" % (items.mean(), *np.percentile(items, [10, 50, 90]).astype(int), items.max(), comp.mean()))
for name, a in (("entry -> fragments and maps in registers", t1 - t0), ("-> first item staged (load, store, barrier)", t2 - t1), ("item loop", t3 - t2), ("epilogue (rope, two staged stores)", t4 - t3), ("whole workgroup", t4 - t0)):
    print("%-48s mean %6.2f us   p10 %6.2f  p50 %6.2f  p90 %6.2f" % (name, us(a).mean(), *us(np.percentile(a, [10, 50, 90]))))
nz = items > 0
print("item loop per item: mean %.2f us" % (us(t3 - t2)[nz] / items[nz]).mean())
