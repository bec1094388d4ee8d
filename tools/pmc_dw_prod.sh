#!/bin/bash
# HBM bytes fetched per launch of the K-major weight-gradient kernel at the production shape's four products
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-dwprod}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_trace --output-format csv -- python3 $R/tools/dbg/dw_prod_shapes.py > $R/gpurun_out/${T}_shapes.txt 2> $R/gpurun_out/${T}_trace.err; echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${T}_fetch --output-format csv -- python3 $R/tools/dbg/dw_prod_shapes.py > /dev/null 2> $R/gpurun_out/${T}_fetch.err; echo "fetch rc=$?"
cd $R && python - <<PY
import csv, glob
rows = []
for fn in glob.glob("gpurun_out/${T}_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] == "FETCH_SIZE" and "gemm8p" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]) * 2 * 1024 / 1e9, r["Grid_Size"] if "Grid_Size" in r else ""))
for d, gb, g in sorted(rows):
    print(f"dispatch {d:4d} grid {g:>8s}: fetched {gb:7.2f} GB")
PY
cat gpurun_out/${T}_shapes.txt
