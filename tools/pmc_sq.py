"""Summarise a rocprofv3 --pmc pass of SQ counters per kernel (sum over dispatches / number of dispatches).
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... -d gpurun_out/pmc_sq1 --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-train-loop
  python tools/pmc_sq.py gpurun_out/pmc_sq1 [name filter ...]"""
import collections, csv, glob, os, sys

d = sys.argv[1]; filt = sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); ids = collections.defaultdict(set)
for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn, newline="")):
        k = r["Kernel_Name"]
        if filt and not any(f in k for f in filt):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); ids[k].add(r.get("Dispatch_Id"))
for k, c in agg.items():
    n = max(len(ids[k]), 1)
    print(k[:100], "x", n)
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    for name, v in sorted(c.items()):
        print(f"   {name:30s} {v / n:14.0f}" + (f"  {v / wc:6.3f} of wave cycles" if wc and name != "SQ_WAVE_CYCLES" else ""))
