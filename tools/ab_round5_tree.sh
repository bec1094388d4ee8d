#!/bin/bash
# same-box A/B of this tree against round 5's final tree (commit e5c06a2, built in the build container into tools/micro/bin/r5tree -- not tracked):
# the default workload (cfg-3) and cfg-2, alternating, un-instrumented step time
R=$GRAFT_REPO_ROOT
FLAGS="--steps 40 --warmup 8 --no-kernel-timing --no-cpu-baseline --no-train-loop --no-extra-legs --no-live-pmc"
for cfg in ${AB_CONFIGS:-cfg3 cfg2}; do
  for rep in 1 2 3; do
    for tree in r5 r6; do
      if [ $tree = r5 ]; then D=$R/tools/micro/bin/r5tree; else D=$R; fi
      (cd $D && python3 bench.py --config $cfg $FLAGS 2>/dev/null) > /tmp/ab5.json || { echo "$tree $cfg failed"; exit 1; }
      python3 -c "import json; d=json.loads(open('/tmp/ab5.json').read().strip().splitlines()[-1]); print('$cfg $tree rep $rep: ms/step', d['ms_per_step'], 'median', d['ms_per_step_stats']['median'])"
    done
  done
done
