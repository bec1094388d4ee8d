set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r5f_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5f_tests.log; tail -5 gpurun_out/r5f_tests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $R/gpurun_out/r5f_bench.json 2> $R/gpurun_out/r5f_bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$R/gpurun_out/r5f_bench.json')); print(d['ms_per_step'], d['ms_per_step_stats'], d['train_loop_ms_per_step']); print(d['ms_per_step_by_phase']); print({k:(v.get('ms_per_step'), v.get('over_resident_step_pct')) for k,v in d['other_configs'].items() if isinstance(v, dict)})"
