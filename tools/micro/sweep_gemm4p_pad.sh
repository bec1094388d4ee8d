#!/bin/bash
# row-stride sweep of tools/micro/gemm4p (G4_PAD = extra elements per operand row): does the operand's leading dimension matter at the step's shapes?
B=$GRAFT_REPO_ROOT/tools/micro/bin/gemm4p
for pad in 0 64 32 192; do
  for shape in "65536 512 2816" "65536 512 1024" "65536 1024 512" "65536 512 512" "65536 2816 512"; do
    G4_PAD=$pad timeout -k 10 120 $B 5 $shape || exit 1
  done
done
