#!/bin/bash
# Host-transport 1x2 rehearsal on the one-GPU box (both ranks share the GPU, messages staged over gloo): rank 1 runs plainly, rank 0
# under rocprofv3 --kernel-trace; tools/overlap_report.py then reads rank 0's trace.  Every process is a plain python program started
# from this shell with its rank in the environment (no launcher, nothing re-execs after the GPU is initialised).
# usage: tools/overlap_run.sh TAG [bench args...]
set -e
TAG=$1; shift
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1 MASTER_PORT=29711 WORLD_SIZE=2 REMORA_BENCH_TRANSPORT=host HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/ovl_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
RANK=1 LOCAL_RANK=1 timeout -k 10 400 python3 bench.py --gpus 2 --no-cpu-baseline "$@" > "$OUT/rank1.log" 2>&1 &
R1=$!
RANK=0 LOCAL_RANK=0 timeout -k 10 400 rocprofv3 --kernel-trace -d "$OUT" -o p --output-format csv -- python3 bench.py --gpus 2 --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/err.log"
wait $R1
python3 tools/overlap_report.py "$OUT"/p_kernel_trace.csv > "$OUT/overlap.txt"
cat "$OUT/overlap.txt"; grep -o '"ms_per_step": [0-9.]*' "$OUT/bench.json"
