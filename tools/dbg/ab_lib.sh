# same-box A/B of two builds of the library: recommendersystem_amd/librsys_hip_{old,new}.so alternately copied over librsys_hip.so
set -e
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2 3; do for v in old new; do cp recommendersystem_amd/librsys_hip_$v.so recommendersystem_amd/librsys_hip.so; python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-train-loop --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d['ms_per_step'], d['ms_per_step_stats']['median'])" $v; done; done
cp recommendersystem_amd/librsys_hip_new.so recommendersystem_amd/librsys_hip.so
