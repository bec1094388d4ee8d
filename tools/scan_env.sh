# Scan one RSYS_* switch over several values on one configuration of bench.py:
#   bash tools/scan_env.sh <config> <grep -E pattern | -> <NAME> <v1> [v2 ...]      (value "-" = unset)
# pattern: lines of the per call-site table (--detail, every call site between HIP events); "-": no table, the UN-instrumented step time
# (three runs per value: mean / median of 40 timed steps each).
R=${GRAFT_REPO_ROOT:-.}; C=$1; PAT=$2; NAME=$3; shift 3
for v in "$@"; do
  if [ "$v" = "-" ]; then unset $NAME; else export $NAME=$v; fi
  if [ "$PAT" = "-" ]; then
    for rep in 1 2 3; do
      python3 $R/bench.py --config $C --steps 40 --warmup 8 --no-kernel-timing --no-cpu-baseline --no-train-loop --no-extra-legs --no-live-pmc 2>/dev/null >/tmp/scan.json
      python3 -c "import json; d=json.loads(open('/tmp/scan.json').read().strip().splitlines()[-1]); print('$NAME=$v   step', d['ms_per_step'], 'median', d['ms_per_step_stats']['median'])"
    done
  else
    python3 $R/bench.py --config $C --steps 20 --warmup 5 --detail --no-cpu-baseline --no-train-loop 2>&1 >/tmp/scan.json | grep -E "$PAT" | sed "s/^/$NAME=$v /" | cut -c1-130
    python3 -c "import json; d=json.loads(open('/tmp/scan.json').read().strip().splitlines()[-1]); print('$NAME=$v   step', d['ms_per_step'], 'median', d['ms_per_step_stats']['median'])"
  fi
done
