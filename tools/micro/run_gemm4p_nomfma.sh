#!/bin/bash
# timing-only variants of tools/micro/gemm4p (gen_gemm4p_asm.py --no-mfma [--a-empty | --b-empty] [--pf]): the loop's request / wait / barrier /
# fragment-read skeleton alone.  "load": N = 512, K = 2816 with 16 ... 256 workgroups and 1 ... 4 tiles each (is the rate a property of the CU or of
# the chip, of the cache or of the memory?); "pf": the A-row touch of --pf, skeleton and real kernel
if [ "$1" = load ]; then
  for b in gemm4p_nomfma gemm4p_nomfma_bempty gemm4p; do
    for m in 2048 8192 16384 32768 65536 131072; do
      timeout -k 10 120 $GRAFT_REPO_ROOT/tools/micro/bin/$b 5 $m 512 2816 | sed "s/^/$b: /" || exit 1
    done
  done
  exit 0
fi
if [ "$1" = pf ]; then
  for shape in "32768 512 2816" "65536 512 2816" "131072 512 2816" "65536 512 1024" "65536 1024 2816" "8192 8192 8192"; do
    for b in gemm4p_nomfma gemm4p_nomfma_pf gemm4p gemm4p_pf; do
      timeout -k 10 120 $GRAFT_REPO_ROOT/tools/micro/bin/$b 5 $shape | sed "s/^/$b: /" || exit 1
    done
  done
  exit 0
fi
if [ "$1" = deep ]; then
  for shape in "2048 2048 1024" "16640 1024 256" "32768 512 2816" "65536 512 2816" "65536 512 1024" "65536 1024 512" "65536 2048 2816" "8192 8192 8192"; do
    for b in gemm4p_nomfma gemm4p_nomfma_deep gemm4p gemm4p_deep; do
      timeout -k 10 120 $GRAFT_REPO_ROOT/tools/micro/bin/$b 5 $shape | sed "s/^/$b: /" || exit 1
    done
  done
  exit 0
fi
if [ "$1" = dma ]; then
  for shape in "65536 512 2816" "65536 512 1024" "65536 1024 512" "65536 2048 2816" "8192 8192 8192"; do
    for b in gemm4p gemm4p_dma_late gemm4p_dma_spread; do
      timeout -k 10 120 $GRAFT_REPO_ROOT/tools/micro/bin/$b 5 $shape | sed "s/^/$b: /" || exit 1
    done
  done
  exit 0
fi
if [ "$1" = nt ]; then
  for shape in "65536 512 2816" "65536 512 1408" "65536 512 1024" "65536 1024 512" "8192 8192 8192"; do
    for b in gemm4p gemm4p_nt_store; do
      timeout -k 10 120 $GRAFT_REPO_ROOT/tools/micro/bin/$b 5 $shape | sed "s/^/$b: /" || exit 1
    done
  done
  exit 0
fi
if [ "$1" = packed ]; then
  for shape in "32768 512 2816" "65536 512 2816" "131072 512 2816" "65536 512 1024"; do
    for b in gemm4p_nomfma gemm4p_nomfma_apacked gemm4p_nomfma_bempty gemm4p_nomfma_apacked_bempty; do
      timeout -k 10 120 $GRAFT_REPO_ROOT/tools/micro/bin/$b 5 $shape | sed "s/^/$b: /" || exit 1
    done
  done
  exit 0
fi
for b in gemm4p_nomfma gemm4p_nomfma_aempty gemm4p_nomfma_bempty; do
  for shape in "65536 512 2816" "65536 512 1408" "65536 2048 2816" "8192 8192 8192"; do
    timeout -k 10 120 $GRAFT_REPO_ROOT/tools/micro/bin/$b 5 $shape | sed "s/^/$b: /" || exit 1
  done
done
