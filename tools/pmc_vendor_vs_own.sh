# MFMA-busy / instruction counters of the vendor GEMM (torch.mm) and of gemm8p on the same shapes (calibration; gpurun -- bash tools/pmc_vendor_vs_own.sh)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for who in vendor own; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_cal_$who --output-format csv -- python3 $R/tools/dbg/${who}_8192.py > /dev/null 2> $R/gpurun_out/pmc_cal_$who.err; echo "$who rc=$?"
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_cal2_$who --output-format csv -- python3 $R/tools/dbg/${who}_8192.py > /dev/null 2> $R/gpurun_out/pmc_cal2_$who.err; echo "$who rc2=$?"
done
cd $R
for who in vendor own; do echo "== $who"; python tools/pmc_mfma_util.py gpurun_out/pmc_cal_$who | grep -i "gemm\|Cijk\|kernel " | cut -c1-170; done
python - <<'PY'
import csv, glob, collections
for who in ("vendor", "own"):
    f = glob.glob(f"gpurun_out/pmc_cal2_{who}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "gemm" in k.lower() or "cijk" in k.lower():
            print(who, k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in d.items()}, "(millions per launch)")
PY
