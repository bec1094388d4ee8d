#!/bin/bash
# builds tools/micro/bin/gemm4p (the library's loop) and its timing-only / measurement variants (gen_gemm4p_asm.py flags) -- run in the build container
cd "$(dirname "$0")/../.." || exit 1
mkdir -p tools/micro/bin
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Irecommendersystem_amd/csrc -Wno-unused-value"
one() {   # name, generator flags...
  local name=$1; shift
  local d=/tmp/gemm4p_$name; mkdir -p $d
  python3 tools/micro/gen_gemm4p_asm.py "$@" $d > /dev/null || exit 1
  hipcc $FLAGS $EXTRA -DGEMM4P_ASM_INC="\"$d/gemm4p_asm.inc\"" tools/micro/gemm4p.hip -o tools/micro/bin/gemm4p_$name 2>&1 | grep -B2 -A6 "error"
}
hipcc $FLAGS tools/micro/gemm4p.hip -o tools/micro/bin/gemm4p 2>&1 | grep -B2 -A6 "error" &
one nomfma --no-mfma &
one nomfma_aempty --no-mfma --a-empty &
one nomfma_bempty --no-mfma --b-empty &
wait
one nomfma_pf --no-mfma --pf &
one pf --pf &
one deep --deep &
one nt_store --nt-store &
one dma_late --dma-at=29,30,31,32 &
one dma_spread --dma-at=10,17,24,31 &
EXTRA=-DGEMM4P_A_PACKED one nomfma_apacked --no-mfma --a-packed &
EXTRA=-DGEMM4P_A_PACKED one nomfma_apacked_bempty --no-mfma --a-packed --b-empty &
one nomfma_deep --no-mfma --deep &
wait
ls -la tools/micro/bin
