# same-box A/B of attention kernel variants: bash tools/ab_attn.sh "ENVVAR" "v1 v2 ..."   (kernel averages from rocprofv3 --stats)
R=$GRAFT_REPO_ROOT
VAR=${1:-RSYS_ATTN_EXP}; VALS=${2:-"0 1"}
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in $VALS; do
export $VAR=$v
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/attn_${v}_$rep --output-format csv -- python3 $R/tools/bench_attn.py 8 > /dev/null 2>&1
echo "$VAR=$v rep $rep"; grep -h "attn_" $R/gpurun_out/attn_${v}_$rep/*/*kernel_stats.csv | awk -F, '{print $1, $2, $4}' | cut -c1-110
done; done
