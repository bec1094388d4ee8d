set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
L=gpurun_out/r5_vendor_gemm_calibration.log
{
echo "# Calibration only -- the product links no vendor GEMM.  Both sides HIP-event timed over back-to-back launches on one box, standard-normal bf16 operands,"
echo "# bf16 output.  (Rounds 3 and 4 timed this library through rsys_op_gemm, which synchronises the device after every call: 150-230 us slower at 8192^3"
echo "# than the same kernel launched back to back -- the last line below.)"
echo "# vendor: python tools/bench_vendor_gemm.py (torch.mm = hipBLASLt)"
python tools/bench_vendor_gemm.py 2>&1 | grep "^M="
echo "# this library: G4_STEP_SHAPES=1 G4_NORMAL=1 tools/micro/bin/gemm4a 20 -- launch_gemm8c (the product's kernel, band order of the output tiles on)"
echo "# and beside it the generated-assembly K loop of tools/micro/gen_gemm4a_asm.py (experiment, not in the library)"
G4_STEP_SHAPES=1 G4_NORMAL=1 ./tools/micro/bin/gemm4a 20 2>&1 | grep "^M="
echo "# vendor, second pass (the box's clock drifts over a run)"
python tools/bench_vendor_gemm.py 2>&1 | grep "^M="
echo "# this library, second pass"
G4_STEP_SHAPES=1 G4_NORMAL=1 ./tools/micro/bin/gemm4a 20 2>&1 | grep "^M="
echo "# row-major tile order (RSYS_GEMM_PATCH=0), the wide shapes"
G4_NORMAL=1 RSYS_GEMM_PATCH=0 ./tools/micro/bin/gemm4a 20 8192 8192 8192 2>&1 | grep "^M="
G4_NORMAL=1 RSYS_GEMM_PATCH=0 ./tools/micro/bin/gemm4a 20 65536 2816 512 2>&1 | grep "^M="
echo "# rsys_op_gemm (device synchronised per call): python tools/bench_gemm.py 8192 8192 8192 0 0"
python tools/bench_gemm.py 8192 8192 8192 0 0 2>&1 | grep "^M="
} > $L 2>&1
cat $L | cut -c1-230
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
