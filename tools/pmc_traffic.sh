#!/bin/bash
# HBM-side traffic per kernel (run on the GPU box through gpurun): two rocprofv3 --pmc passes of their own (FETCH_SIZE, WRITE_SIZE
# cannot share a pass: MI355X_MICROARCH.md, rocprofv3 PMC slots), no tracing options next to --pmc.
# usage: tools/pmc_traffic.sh TAG [bench args...]
set -e
TAG=$1; shift
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=$PWD/gpurun_out/pmc_${TAG}_$C
  rm -rf "$OUT"; mkdir -p "$OUT"
  timeout -s KILL 500 rocprofv3 --pmc $C -d "$OUT" -o p --output-format csv -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/err.log"
  ls "$OUT" | head
done
