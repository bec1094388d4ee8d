# same-box A/B of the weight-gradient side stream: RSYS_SIDE_STREAM unset / 1 (join behind the paired dx GEMM) / 2 (deferred joins)
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for mode in 0 1 2; do
    if [ $mode = 0 ]; then unset RSYS_SIDE_STREAM; else export RSYS_SIDE_STREAM=$mode; fi
    python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train-loop --no-kernel-timing > /tmp/ab.json 2>/tmp/ab.err || { echo "mode $mode failed"; tail -3 /tmp/ab.err; exit 1; }
    python3 -c "
import json;d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]);print('side_stream=$mode rep $rep', d['ms_per_step'], d['ms_per_step_stats']['median'] if 'ms_per_step_stats' in d else '', d['losses'])"
  done
done
