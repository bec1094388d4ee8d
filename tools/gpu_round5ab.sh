set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for extra in "" "--zero1" "--split-table-reduce"; do
timeout -k 10 300 python bench.py --rehearse-comm $extra --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/r5ab.json 2> gpurun_out/r5ab.err; echo "rehearse '$extra' rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r5ab.json')); print(d['ms_per_step'], d['ms_per_step_stats']['median'], d.get('allreduce_exposed_ms_per_step'), json.dumps(d['comm'])[:400], d['config']['parallelism'], d['roofline']['frac'])" || tail -5 gpurun_out/r5ab.err
done
