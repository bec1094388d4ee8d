# One GPU: the data-parallel step with its RCCL collectives issued at a forced world of 1 (bench.py --rehearse-comm), in the three
# gradient-reduction modes; prints step time, exposed all-reduce and the bucket schedule of each.
R=${GRAFT_REPO_ROOT:-.}
for extra in "" "--zero1" "--split-table-reduce"; do
  timeout -k 10 300 python3 $R/bench.py --rehearse-comm $extra --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > /tmp/rehearse.json 2> /tmp/rehearse.err; echo "rehearse '$extra' rc=$?"
  python3 -c "
import json; d=json.loads(open('/tmp/rehearse.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_stats']['median'], 'exposed', d.get('allreduce_exposed_ms_per_step'), 'replicas', d.get('replicas_consistent'), json.dumps(d['comm'])[:600], d['config']['parallelism'])" || tail -5 /tmp/rehearse.err
done
