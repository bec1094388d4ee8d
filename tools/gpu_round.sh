# One call on the GPU box, assembled from named steps:   gpurun -- bash tools/gpu_round.sh <tag> <step> [<step> ...]
# Everything a step writes goes to gpurun_out/<tag>_*; copy what is to be judged into profiles/ afterwards.  Steps run in order and the
# script stops at the first failing one (a GPU step that was killed must not be followed by another one).
#
#   tests[=<-k expression>]      pytest -m gpu (whole suite, or the tests the expression selects), ONE process
#   bench[=<config>[,<dtype>]]   the driver's line (default flags) for cfg3, or `--config <config>` with 20 steps / 5 warm-up
#   detail[=<config>[,<dtype>]]  per call-site table (bench.py --detail), 20 steps
#   prof[=<config>]              rocprofv3 --kernel-trace --stats of the bench + the steady-state per-step table (tools/trace_steady.py)
#   pmc[=<config>]               HBM bytes per launch: separate FETCH_SIZE / WRITE_SIZE passes + tools/pmc_traffic.py
#   mfma[=<config>]              MFMA busy per kernel: one --pmc pass + tools/pmc_mfma_util.py
#   ab=<NAME>:<VALUE>[:reps]     same-box A/B of one RSYS_* switch on the default bench (tools/ab_env.sh)
#   abdetail=<NAME>:<VALUE>[,<config>]  the per call-site table with the switch off and on (one run each)
#   vendor                       hipBLASLt calibration of the GEMM shapes (tools/bench_vendor_gemm.py)
#   py=<script>[,args...]        python3 tools/<script> args   (micro benchmarks: bench_attn.py, bench_gemm.py, ...)
#   sh=<script>[,args...]        bash tools/<script> args
set -o pipefail
R=$GRAFT_REPO_ROOT
T=${1:?tag}; shift
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
QUIET="--no-cpu-baseline --no-train-loop --no-extra-legs --no-live-pmc"
cfgflags() {   # "<config>[,<dtype>]" -> bench.py flags
  local c=${1%%,*} d=""
  [[ "$1" == *,* ]] && d=${1#*,}
  echo "--config ${c:-cfg3}${d:+ --dtype $d}"
}
for step in "$@"; do
  name=${step%%=*}; arg=""; [[ "$step" == *=* ]] && arg=${step#*=}
  c=${arg%%,*}; c=${c:-cfg3}
  t0=$(date +%s)
  case $name in
    tests)
      if [ -n "$arg" ]; then
        (cd $R && timeout -k 10 1100 python -m pytest tests -m gpu -q -x -s -k "$arg") > $O/${T}_tests.log 2>&1; rc=$?
      else
        (cd $R && timeout -k 10 1100 python -m pytest tests -m gpu -q -x) > $O/${T}_tests.log 2>&1; rc=$?
      fi
      echo "pytest rc=$rc" >> $O/${T}_tests.log; tail -6 $O/${T}_tests.log | cut -c1-220 ;;
    bench)
      if [ -z "$arg" ]; then
        timeout -k 10 900 python3 $R/bench.py > $O/${T}_bench_cfg3.json 2> $O/${T}_bench_cfg3.err; rc=$?
      else
        timeout -k 10 600 python3 $R/bench.py $(cfgflags "$arg") --steps 20 --warmup 5 --no-cpu-baseline --no-train-loop > $O/${T}_bench_${c}.json 2> $O/${T}_bench_${c}.err; rc=$?
      fi
      python3 - $O/${T}_bench_${c}.json <<'EOF'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "median", d.get("ms_per_step_stats", {}).get("median"), "value", d["value"], "step_mfma_frac", d["step_mfma_frac"])
print("roofline", json.dumps(d["roofline"])[:400])
print("phases", d.get("ms_per_step_by_phase"))
print("hbm", {k: (v["GBps"], v["ms_per_step"]) for k, v in d.get("hbm_kernels", {}).items()})
print("other", {k: (v.get("ms_per_step"), v.get("error")) for k, v in d.get("other_configs", {}).items() if isinstance(v, dict)})
EOF
      ;;
    detail)
      timeout -k 10 600 python3 $R/bench.py $(cfgflags "$arg") --steps 20 --warmup 5 --detail --no-cpu-baseline --no-train-loop > $O/${T}_detail_${c}.json 2> $O/${T}_bench_detail_${c}.txt; rc=$?
      head -40 $O/${T}_bench_detail_${c}.txt | cut -c1-120 ;;
    prof)
      rocprofv3 --kernel-trace --stats -d $O/prof_${T}_${c} --output-format csv -- python3 $R/bench.py $(cfgflags "$arg") --steps 20 --warmup 5 $QUIET > $O/${T}_prof_${c}.json 2> $O/${T}_prof_${c}.err; rc=$?
      (cd $R && python3 tools/trace_steady.py $O/prof_${T}_${c} > $O/${T}_${c}_steady_state_per_step.txt) && head -24 $O/${T}_${c}_steady_state_per_step.txt | cut -c1-170
      cp $(find $O/prof_${T}_${c} -name "*kernel_stats.csv" | head -1) $O/${T}_${c}_kernel_stats.csv
      find $O/prof_${T}_${c} -name "*kernel_trace.csv" -delete ;;
    pmc)
      rc=0
      for ctr in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $ctr -d $O/pmc_${ctr}_${T}_${c} --output-format csv -- python3 $R/bench.py $(cfgflags "$arg") --steps 3 --warmup 1 --no-kernel-timing $QUIET > /dev/null 2> $O/${T}_pmc_${ctr}_${c}.err || rc=$?
      done
      (cd $R && python3 tools/pmc_traffic.py $O/pmc_FETCH_SIZE_${T}_${c} $O/pmc_WRITE_SIZE_${T}_${c} $O/${T}_${c}_pmc_traffic.json | head -14)
      rm -rf $O/pmc_FETCH_SIZE_${T}_${c} $O/pmc_WRITE_SIZE_${T}_${c} ;;
    mfma)
      rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma_${T}_${c} --output-format csv -- python3 $R/bench.py $(cfgflags "$arg") --steps 3 --warmup 1 --no-kernel-timing $QUIET > /dev/null 2> $O/${T}_pmc_mfma_${c}.err; rc=$?
      (cd $R && python3 tools/pmc_mfma_util.py $O/pmc_mfma_${T}_${c} > $O/${T}_${c}_pmc_mfma_util.txt) && head -16 $O/${T}_${c}_pmc_mfma_util.txt | cut -c1-170
      rm -rf $O/pmc_mfma_${T}_${c} ;;
    ab)
      IFS=: read -r n v reps <<< "$arg"
      bash $R/tools/ab_env.sh $n $v ${reps:-2} > $O/${T}_ab_${n}.log 2>&1; rc=$?; cut -c1-200 $O/${T}_ab_${n}.log ;;
    abdetail)
      IFS=: read -r n v <<< "${arg%%,*}"; c=cfg3; [[ "$arg" == *,* ]] && c=${arg#*,}
      rc=0
      for arm in off on; do
        if [ $arm = on ]; then export $n=$v; else unset $n; fi
        timeout -k 10 600 python3 $R/bench.py --config $c --steps 20 --warmup 5 --detail --no-cpu-baseline --no-train-loop > $O/${T}_abdetail_${n}_${arm}.json 2> $O/${T}_abdetail_${n}_${c}_${arm}.txt || rc=$?
        python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('$n=$v $arm: ms/step', d['ms_per_step'], 'median', d['ms_per_step_stats']['median'])" $O/${T}_abdetail_${n}_${arm}.json
      done
      unset $n
      paste <(head -44 $O/${T}_abdetail_${n}_${c}_off.txt | cut -c1-62) <(head -44 $O/${T}_abdetail_${n}_${c}_on.txt | cut -c1-62) ;;
    vendor)
      timeout -k 10 600 python3 $R/tools/bench_vendor_gemm.py > $O/${T}_vendor_gemm_calibration.log 2>&1; rc=$?; tail -14 $O/${T}_vendor_gemm_calibration.log | cut -c1-200 ;;
    py)
      IFS=, read -r -a a <<< "$arg"
      (cd $R && timeout -k 10 900 python3 tools/${a[0]} "${a[@]:1}") > $O/${T}_${a[0]##*/}.log 2>&1; rc=$?; tail -30 $O/${T}_${a[0]##*/}.log | cut -c1-220 ;;
    sh)
      IFS=, read -r -a a <<< "$arg"
      (cd $R && timeout -k 10 900 bash tools/${a[0]} "${a[@]:1}") > $O/${T}_${a[0]##*/}.log 2>&1; rc=$?; tail -30 $O/${T}_${a[0]##*/}.log | cut -c1-220 ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
  echo "== $step rc=$rc in $(( $(date +%s) - t0 )) s"
  [ $rc -ne 0 ] && { tail -5 $O/${T}_*${c}*.err 2>/dev/null | cut -c1-300; exit $rc; }
done
exit 0
