#!/bin/bash
# Premise test for CU-partitioned concurrency: two processes, each masked to half of the CUs of every XCD (HSA_CU_MASK),
# each running the cfg-3 step on HALF of the benchmark batch (32 rows), against one process on the whole chip.
# If 2 x 32 rows finish sooner side by side than 64 rows on the whole chip, HBM-bound and MFMA-bound phases of the two
# halves overlap.   bash tools/ab_cu_mask.sh   (on the GPU box)
set -o pipefail
OUT=gpurun_out/ab_cu_mask; mkdir -p $OUT
B="python bench.py --warmup 10 --no-cpu-baseline --no-kernel-timing --no-train-loop"
pick() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["ms_per_step_stats"]["median"], d["config"]["rows_per_gpu"])
PY
}
$B --steps 60 --rows 64 > $OUT/full64.log 2>$OUT/full64.err && pick $OUT/full64.log
$B --steps 60 --rows 32 > $OUT/full32.log 2>$OUT/full32.err && pick $OUT/full32.log
HSA_CU_MASK=0:0-127 $B --steps 60 --rows 32 > $OUT/half32_alone.log 2>$OUT/half32_alone.err && pick $OUT/half32_alone.log
HSA_CU_MASK=0:0-127 $B --steps 300 --rows 32 > $OUT/half32_a.log 2>$OUT/half32_a.err &
PA=$!
HSA_CU_MASK=0:128-255 $B --steps 300 --rows 32 > $OUT/half32_b.log 2>$OUT/half32_b.err &
PB=$!
wait $PA; wait $PB
pick $OUT/half32_a.log; pick $OUT/half32_b.log
# unmasked pair for comparison (the dispatcher's own sharing)
$B --steps 300 --rows 32 > $OUT/pair32_a.log 2>$OUT/pair32_a.err &
PA=$!
$B --steps 300 --rows 32 > $OUT/pair32_b.log 2>$OUT/pair32_b.err &
PB=$!
wait $PA; wait $PB
pick $OUT/pair32_a.log; pick $OUT/pair32_b.log
