# LDS-array activity and bank conflicts of every kernel of the cfg-3 step (one rocprofv3 --pmc pass): bash tools/pmc_lds.sh <tag>
R=$GRAFT_REPO_ROOT
T=${1:-r4}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_lds_$T --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-train-loop > /dev/null 2> $R/gpurun_out/pmc_lds_$T.err; echo rc=$?
cd $R && python tools/pmc_sq.py gpurun_out/pmc_lds_$T gemm attn_ > gpurun_out/${T}_pmc_lds.txt
