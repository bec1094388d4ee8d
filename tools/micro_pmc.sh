#!/bin/bash
# bash tools/micro_pmc.sh <binary under tools/micro/bin> <args...>: HBM traffic per kernel of a prebuilt micro-benchmark (separate FETCH_SIZE / WRITE_SIZE
# passes, MI355X_MICROARCH.md's correction: bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB), the program directly behind `--`
B=$GRAFT_REPO_ROOT/tools/micro/bin/$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/micro_pmc
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/$ctr
  timeout -k 10 300 rocprofv3 --pmc $ctr -d $O/$ctr --output-format csv -- $B "$@" > /dev/null 2> $O.$ctr.err || { tail -5 $O.$ctr.err; exit 1; }
done
python3 - $O <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for i, ctr in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    for f in glob.glob(f"{o}/{ctr}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr: continue
            k = r["Kernel_Name"][:70]
            tot[k][i] += float(r["Counter_Value"])
            if i == 0: tot[k][2] += 1
for k, (fe, wr, n) in sorted(tot.items()):
    n = max(n, 1)
    print(f"{k:72s} launches {n:3d}  fetch {2 * fe * 1024 / n / 1e6:9.1f} MB  write {wr * 1024 / n / 1e6:9.1f} MB per launch")
PY
rm -rf $O
