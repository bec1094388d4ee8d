R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-train-loop"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_sq1 --output-format csv -- $B > /dev/null 2> $R/gpurun_out/pmc_sq1.err; echo rc=$?
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM -d $R/gpurun_out/pmc_sq2 --output-format csv -- $B > /dev/null 2> $R/gpurun_out/pmc_sq2.err; echo rc=$?
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_CVT SQ_LEVEL_WAVES -d $R/gpurun_out/pmc_sq3 --output-format csv -- $B > /dev/null 2> $R/gpurun_out/pmc_sq3.err; echo rc=$?
cd $R
for i in 1 2 3; do python tools/pmc_sq.py gpurun_out/pmc_sq$i attn_ > gpurun_out/pmc_sq$i.txt; done
cat gpurun_out/pmc_sq1.txt gpurun_out/pmc_sq2.txt gpurun_out/pmc_sq3.txt
