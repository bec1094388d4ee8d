set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for shape in "8192 8192 8192" "4096 4096 8192"; do
tag=$(echo $shape | tr ' ' 'x')
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/r5k_pmc_$tag --output-format csv -- $R/tools/micro/bin/gemm4a 2 $shape > $R/gpurun_out/r5k_pmc_$tag.log 2>&1
echo "== $shape"; python3 $R/tools/pmc_mfma_util.py $R/gpurun_out/r5k_pmc_$tag | cut -c1-160
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $R/gpurun_out/r5k_sq_$tag --output-format csv -- $R/tools/micro/bin/gemm4a 2 $shape > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob("$R/gpurun_out/r5k_sq_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
for k,c in agg.items():
    wc=c.get("SQ_WAVE_CYCLES",1)
    print(k, {n: round(v/wc,3) for n,v in sorted(c.items()) if n!="SQ_WAVE_CYCLES"}, "wave_cycles", int(wc))
PY
done
