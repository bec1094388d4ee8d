# rocprofv3 evidence for the fp8 trunk (gpurun -- bash tools/prof_fp8.sh): kernel-trace summary at the production shape and at cfg-3, MFMA-busy counters at cfg-3
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fp8_prod --output-format csv -- python3 $R/bench.py --config prod --dtype fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-train-loop --no-kernel-timing > $R/gpurun_out/prof_fp8_prod.json 2> $R/gpurun_out/prof_fp8_prod.err; echo "prod rc=$?"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fp8_cfg3 --output-format csv -- python3 $R/bench.py --dtype fp8 --steps 20 --warmup 5 --no-cpu-baseline --no-train-loop --no-kernel-timing > $R/gpurun_out/prof_fp8_cfg3.json 2> $R/gpurun_out/prof_fp8_cfg3.err; echo "cfg3 rc=$?"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_mfma_fp8 --output-format csv -- python3 $R/bench.py --dtype fp8 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-train-loop > /dev/null 2> $R/gpurun_out/pmc_mfma_fp8.err; echo "pmc rc=$?"
cd $R && python tools/pmc_mfma_util.py gpurun_out/pmc_mfma_fp8 > gpurun_out/r3_fp8_pmc_mfma_util.txt; head -14 gpurun_out/r3_fp8_pmc_mfma_util.txt | cut -c1-150
