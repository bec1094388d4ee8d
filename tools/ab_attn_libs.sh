# same-box A/B of attention kernel BUILDS (timing-only variants under tools/micro/bin): bash tools/ab_attn_libs.sh name1 name2 ...
# ("-" = the shipped library); kernel averages from rocprofv3 --stats of tools/bench_attn.py
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in "$@"; do
if [ "$v" = "-" ]; then unset RSYS_LIB_PATH; else export RSYS_LIB_PATH=$R/tools/micro/bin/librsys_$v.so; fi
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/attnlib_${v}_$rep --output-format csv -- python3 $R/tools/bench_attn.py 8 > /dev/null 2>&1
echo "lib=$v rep $rep"; grep -h "attn_fwd\|attn_bwd" $R/gpurun_out/attnlib_${v}_$rep/*/*kernel_stats.csv | awk -F, '{print $1, $2, $4}' | cut -c1-110
done; done
