#!/bin/bash
# which hipBLASLt kernels torch.mm picks for the calibration shapes (their names encode macro tile, depth, LDS buffering, wave layout): calibration only
O=$GRAFT_REPO_ROOT/gpurun_out/vendor_names
cd /tmp && export TMPDIR=/tmp
rm -rf $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_vendor_gemm.py > /dev/null 2> $O.err || { tail -5 $O.err; exit 1; }
python3 - $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "Cijk" in n or "gemm" in n.lower():
            print(r["Calls"], r["AverageNs"], n[:400])
PY
rm -rf $O
