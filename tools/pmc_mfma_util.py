"""MFMA utilisation per kernel from one rocprofv3 --pmc pass over a bench run:
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_mfma --output-format csv -- python3 bench.py ...
  python tools/pmc_mfma_util.py gpurun_out/pmc_mfma
util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs,
MI355X_MICROARCH.md "DVFS give-back"); MFMA_BUSY counts cycles per SIMD (16 per v_mfma_f32_16x16x32_bf16)."""
import collections, csv, glob, os, sys

d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); ids = collections.defaultdict(set)
for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn, newline="")):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); ids[k].add(r.get("Dispatch_Id"))
rows = []
for k, c in agg.items():
    n = max(len(ids[k]), 1)
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    rows.append((gui, k, n, busy, c.get("SQ_INSTS_MFMA", 0.0), c.get("SQ_BUSY_CYCLES", 0.0)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{'kernel':72s} {'launches':>8s} {'share':>6s} {'kcycles/launch':>14s} {'MFMA busy/SIMD-cycle':>20s} {'MFMA insts/launch':>18s}")
for gui, k, n, busy, insts, sqb in rows:
    if gui / max(tot, 1) < 0.002:
        continue
    print(f"{k[:72]:72s} {n:8d} {gui / tot:6.3f} {gui / n / 1e3:14.1f} {busy / max(gui * 1024, 1):20.3f} {insts / n:18.0f}")
