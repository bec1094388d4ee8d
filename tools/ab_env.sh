# same-box A/B of an environment switch on the default bench: bash tools/ab_env.sh NAME VALUE [reps]
R=$GRAFT_REPO_ROOT; NAME=$1; VAL=$2; REPS=${3:-2}
for rep in $(seq 1 $REPS); do
  for arm in off on; do
    if [ $arm = on ]; then export $NAME=$VAL; else unset $NAME; fi
    python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train-loop --no-live-pmc > /tmp/ab.json 2>/tmp/ab.err || { echo "$arm failed"; tail -3 /tmp/ab.err; exit 1; }
    python3 -c "
import json;d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]);g=d['gemm_variants'];print('$NAME=$VAL $arm rep $rep', d['ms_per_step'], d['ms_per_step_stats']['median'], {k:(v['ms_per_step'],v['tflops']) for k,v in g.items()})"
  done
done
