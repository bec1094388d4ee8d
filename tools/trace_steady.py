"""Per-step kernel accounting of the STEADY state from a `rocprofv3 --kernel-trace` run of bench.py: launches and microseconds per
optimizer step for every kernel, counted between AdamW launches after the warm-up (rocprofv3's --stats summary mixes in the
one-off work of model creation: ~200 allocation-time zero fills of 20 GB, the metadata table, plan uploads).

    python tools/trace_steady.py gpurun_out/prof_r3a [skip_steps] > profiles/r3a_steady_state_per_step.txt
"""
import collections, csv, glob, os, sys

d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 5
fn = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
first, last = ad[skip - 1], ad[-1]
steps = len(ad) - skip
seg = rows[first + 1:last + 1]
tot, cnt = collections.Counter(), collections.Counter()
for r in seg:
    n = r["Kernel_Name"]
    tot[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[n] += 1
span = (int(rows[last]["End_Timestamp"]) - int(rows[first]["End_Timestamp"])) / steps / 1e6
busy = sum(tot.values()) / steps / 1e6
print(f"# {os.path.basename(fn)}: {steps} steady-state steps (after {skip}); {span:.3f} ms/step between AdamW launches, {busy:.3f} ms/step inside kernels "
      f"({span - busy:.3f} ms of gaps), {sum(cnt.values()) / steps:.1f} launches/step")
print(f"# {'us/step':>9s} {'launches/step':>13s} {'us/launch':>10s}  kernel")
for n, t in tot.most_common():
    print(f"{t / steps / 1e3:11.1f} {cnt[n] / steps:13.1f} {t / cnt[n] / 1e3:10.1f}  {n[:150]}")
