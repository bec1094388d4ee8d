#!/bin/bash
# bash tools/run_micro.sh <binary under tools/micro/bin> [args...]: a prebuilt micro-benchmark under a timeout (the binary travels with the snapshot)
B=$1; shift
timeout -k 10 ${MICRO_TIMEOUT:-240} "$GRAFT_REPO_ROOT/tools/micro/bin/$B" "$@"
