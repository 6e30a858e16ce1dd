#!/bin/bash
# One rocprofv3 --pmc pass of the bench (run on the GPU box through gpurun).  usage: tools/pmc_pass.sh TAG "COUNTER ..." [bench args...]
set -e
TAG=$1; CTRS=$2; shift; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -s KILL 500 rocprofv3 --pmc $CTRS -d "$OUT" -o p --output-format csv -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/err.log"
ls "$OUT" | head
