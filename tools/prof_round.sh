set -e
R=$GRAFT_REPO_ROOT
python -m pytest $R/tests -m gpu -q > $R/gpurun_out/r3a_tests_full.log 2>&1; echo "pytest rc=$?" >> $R/gpurun_out/r3a_tests_full.log; tail -4 $R/gpurun_out/r3a_tests_full.log
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/r3a_bench_cfg3.json 2> $R/gpurun_out/r3a_bench_cfg3.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r3a --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-loop > $R/gpurun_out/prof_r3a.json 2> $R/gpurun_out/prof_r3a.err; echo "prof rc=$?"
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_r3a --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-train-loop > $R/gpurun_out/pmc_fetch_r3a.json 2> $R/gpurun_out/pmc_fetch_r3a.err; echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_r3a --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-train-loop > $R/gpurun_out/pmc_write_r3a.json 2> $R/gpurun_out/pmc_write_r3a.err; echo "pmc write rc=$?"
cd $R && python tools/pmc_traffic.py gpurun_out/pmc_fetch_r3a gpurun_out/pmc_write_r3a gpurun_out/r3a_pmc_traffic.json | head -8
find gpurun_out/prof_r3a -name "*kernel_stats.csv" | head -2
cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_mfma --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-train-loop > /dev/null 2> $R/gpurun_out/pmc_mfma.err; echo "pmc mfma rc=$?"
cd $R && python tools/pmc_mfma_util.py gpurun_out/pmc_mfma > gpurun_out/pmc_mfma_util.txt; head -12 gpurun_out/pmc_mfma_util.txt | cut -c1-150
