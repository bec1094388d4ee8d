# Scan BUILDS of the library (compile-time variants under tools/micro/bin/librsys_<name>.so; "-" = the shipped one) on one configuration:
#   bash tools/scan_libs.sh <config> <grep -E pattern | -> <name1> [name2 ...]      (pattern / "-" as tools/scan_env.sh)
R=${GRAFT_REPO_ROOT:-.}; C=$1; PAT=$2; shift 2
for v in "$@"; do
  if [ "$v" = "-" ]; then unset RSYS_LIB_PATH; else export RSYS_LIB_PATH=$R/tools/micro/bin/librsys_$v.so; fi
  if [ "$PAT" = "-" ]; then
    for rep in 1 2 3; do
      python3 $R/bench.py --config $C --steps 40 --warmup 8 --no-kernel-timing --no-cpu-baseline --no-train-loop --no-extra-legs --no-live-pmc 2>/dev/null >/tmp/scan.json
      python3 -c "import json; d=json.loads(open('/tmp/scan.json').read().strip().splitlines()[-1]); print('lib=$v   step', d['ms_per_step'], 'median', d['ms_per_step_stats']['median'])"
    done
  else
    python3 $R/bench.py --config $C --steps 20 --warmup 5 --detail --no-cpu-baseline --no-train-loop 2>&1 >/tmp/scan.json | grep -E "$PAT" | sed "s/^/lib=$v /" | cut -c1-130
  fi
done
