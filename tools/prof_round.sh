set -e
R=$GRAFT_REPO_ROOT
python -m pytest $R/tests -m gpu -q > $R/gpurun_out/r2_tests_full.log 2>&1; echo "pytest rc=$?" >> $R/gpurun_out/r2_tests_full.log; tail -4 $R/gpurun_out/r2_tests_full.log
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/r2b_bench_cfg3.json 2> $R/gpurun_out/r2b_bench_cfg3.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2b --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-loop > $R/gpurun_out/prof_r2b.json 2> $R/gpurun_out/prof_r2b.err; echo "prof rc=$?"
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_r2b --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-train-loop > $R/gpurun_out/pmc_fetch_r2b.json 2> $R/gpurun_out/pmc_fetch_r2b.err; echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_r2b --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-train-loop > $R/gpurun_out/pmc_write_r2b.json 2> $R/gpurun_out/pmc_write_r2b.err; echo "pmc write rc=$?"
cd $R && python tools/pmc_traffic.py gpurun_out/pmc_fetch_r2b gpurun_out/pmc_write_r2b gpurun_out/r2b_pmc_traffic.json | head -8
find gpurun_out/prof_r2b -name "*kernel_stats.csv" | head -2
