#!/bin/bash
# Kernel-trace profile of the bench command (run on the GPU box through gpurun): writes gpurun_out/prof_$1/
# usage: tools/profile.sh TAG [bench args...]
set -e
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT" -o p --output-format csv -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/err.log"
ls -R "$OUT" | head -20
