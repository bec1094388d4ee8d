# where the wall time of a one-user inference call goes: kernel time (rocprofv3 --kernel-trace --stats) against the wall time per call
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_infer --output-format csv -- python3 $R/tools/bench_infer.py cfg3 > $R/gpurun_out/prof_infer.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_infer/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# calls are separated by the tilemap kernel (one per forward)
starts = [i for i, e in enumerate(ev) if "attn_tilemap" in e[2]]
import collections
per = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = ev[a:b]
    per.append((len(seg), sum(e - s for s, e, _ in seg) / 1e3, (seg[-1][1] - seg[0][0]) / 1e3))
# group by kernel count signature -> first 46 calls are rows=1 (2 tasks x (3+20) x 2 variants)
for i in (5, 30, 60, 100, 150, 200, 250):
    if i < len(per):
        print(f"call {i}: {per[i][0]} kernels, kernel time {per[i][1]:.1f} us, first-to-last span {per[i][2]:.1f} us")
PY
