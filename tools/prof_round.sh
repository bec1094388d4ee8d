# One measurement round on the GPU box (gpurun -- bash tools/prof_round.sh <tag>): the GPU test suite, the PMC traffic passes (first:
# the bench line quotes their summary), the driver line, the rocprofv3 kernel-trace summary of the same command, the MFMA-busy counters.
set -e
R=$GRAFT_REPO_ROOT
T=${1:-r3a}
python -m pytest $R/tests -m gpu -q > $R/gpurun_out/${T}_tests_full.log 2>&1; echo "pytest rc=$?" >> $R/gpurun_out/${T}_tests_full.log; tail -4 $R/gpurun_out/${T}_tests_full.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch_$T --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-train-loop > $R/gpurun_out/pmc_fetch_$T.json 2> $R/gpurun_out/pmc_fetch_$T.err; echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write_$T --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-train-loop > $R/gpurun_out/pmc_write_$T.json 2> $R/gpurun_out/pmc_write_$T.err; echo "pmc write rc=$?"
cd $R && python tools/pmc_traffic.py gpurun_out/pmc_fetch_$T gpurun_out/pmc_write_$T gpurun_out/${T}_pmc_traffic.json | head -8
cp gpurun_out/${T}_pmc_traffic.json profiles/${T}_pmc_traffic.json
cd /tmp
python3 $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/${T}_bench_cfg3.json 2> $R/gpurun_out/${T}_bench_cfg3.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$T --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-loop --no-extra-legs --no-live-pmc > $R/gpurun_out/prof_$T.json 2> $R/gpurun_out/prof_$T.err; echo "prof rc=$?"
find $R/gpurun_out/prof_$T -name "*kernel_stats.csv" | head -2
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_mfma_$T --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-train-loop > /dev/null 2> $R/gpurun_out/pmc_mfma_$T.err; echo "pmc mfma rc=$?"
cd $R && python tools/pmc_mfma_util.py gpurun_out/pmc_mfma_$T > gpurun_out/${T}_pmc_mfma_util.txt; head -12 gpurun_out/${T}_pmc_mfma_util.txt | cut -c1-150
cd $R && python tools/trace_steady.py gpurun_out/prof_$T > gpurun_out/${T}_steady_state_per_step.txt; head -14 gpurun_out/${T}_steady_state_per_step.txt | cut -c1-150
cd /tmp && python3 $R/bench.py --steps 20 --warmup 5 --detail --no-cpu-baseline --no-train-loop > /dev/null 2> $R/gpurun_out/${T}_bench_detail.txt; echo "detail rc=$?"; head -12 $R/gpurun_out/${T}_bench_detail.txt | cut -c1-120
