#!/bin/bash
# Issue-side counters per kernel (run on the GPU box through gpurun): one rocprofv3 --pmc pass of eight SQ counters + GRBM_GUI_ACTIVE,
# no tracing options next to --pmc.   usage: tools/pmc_sq.sh TAG [bench args...]
set -e
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_${TAG}_SQ
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE -d "$OUT" -o p --output-format csv -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/err.log"
ls "$OUT" | head
