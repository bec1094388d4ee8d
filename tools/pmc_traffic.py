"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command) into
profiles/<name>.json: HBM bytes per launch for every kernel.

  rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing
  rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r1d_pmc_traffic.json

Correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB and on gfx950 FETCH_SIZE counts half
of a coalesced read: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import csv, glob, json, os, sys


def collect(d, counter):
    agg = {}
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(fn, newline="") as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != counter:
                    continue
                a = agg.setdefault(r["Kernel_Name"], {"sum": 0.0, "ids": set()})
                a["sum"] += float(r["Counter_Value"])
                a["ids"].add(r.get("Dispatch_Id") or r.get("Correlation_Id"))
    return {k: (v["sum"], len(v["ids"])) for k, v in agg.items()}


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing",
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md HBM section (gfx950 FETCH_SIZE counts half of a coalesced read; units are KiB)",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        fs, fn = fetch.get(k, (0.0, 0)); ws, wn = write.get(k, (0.0, 0))
        n = max(fn, wn, 1)
        out["kernels"][k] = {"launches": n, "fetch_size_kib_per_launch": fs / max(fn, 1), "write_size_kib_per_launch": ws / max(wn, 1),
                             "hbm_bytes_per_launch": (2 * fs / max(fn, 1) + ws / max(wn, 1)) * 1024}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{v['launches']:6d} x {v['hbm_bytes_per_launch']/1e6:10.1f} MB  {k[:90]}")


if __name__ == "__main__":
    main()
