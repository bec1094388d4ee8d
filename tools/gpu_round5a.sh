set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
./tools/micro/bin/pk_f32_crossed_probe 2048 4096 4 > gpurun_out/r5a_pk_probe.log 2>&1; echo "probe rc=$?"; cat gpurun_out/r5a_pk_probe.log
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r5a_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5a_tests.log; tail -5 gpurun_out/r5a_tests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $R/gpurun_out/r5a_bench.json 2> $R/gpurun_out/r5a_bench.err; echo "bench rc=$?"; cut -c1-600 $R/gpurun_out/r5a_bench.json
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r5a_attn --output-format csv -- python3 $R/tools/bench_attn.py 8 > /dev/null 2>&1; grep -h "attn_" $R/gpurun_out/r5a_attn/*/*kernel_stats.csv | awk -F, '{print $1, $2, $4}' | cut -c1-140
