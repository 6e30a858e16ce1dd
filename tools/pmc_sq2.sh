#!/bin/bash
# LDS / memory-instruction counters per kernel (second SQ pass; see tools/pmc_sq.sh).   usage: tools/pmc_sq2.sh TAG [bench args...]
set -e
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_${TAG}_SQ2
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE -d "$OUT" -o p --output-format csv -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/err.log"
ls "$OUT" | head
