#!/usr/bin/env python3
"""Per-kernel issue-side picture from the pass of tools/pmc_sq.sh.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
summed over waves (MI355X_MICROARCH.md, per-instruction table): per wave, ACTIVE_INST_ANY + WAIT_INST_ANY + WAIT_ANY ~ WAVE_CYCLES.
usage: pmc_sq_summary.py TAG"""
import csv, collections, glob, sys
tag = sys.argv[1]
path = glob.glob(f"gpurun_out/pmc_{tag}_SQ/*counter_collection.csv")[0]
val = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int); dur = collections.defaultdict(float)
seen = set()
for row in csv.DictReader(open(path)):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
    val[k][row["Counter_Name"]] += float(row["Counter_Value"])
    key = (row["Dispatch_Id"], k)
    if key not in seen:
        seen.add(key); cnt[k] += 1; dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
print(f"{'kernel':36s} {'n':>4s} {'ms':>7s} {'waves':>9s} {'VALU/wave':>9s} {'SALU/wave':>9s} {'valu_act':>8s} {'any_act':>8s} {'wait_inst':>9s} {'wait_any':>8s} {'simd_valu_busy':>14s}")
for k in sorted(val, key=lambda k: -dur[k])[:22]:
    v = val[k]; n = cnt[k]; w = v["SQ_WAVES"]
    if w == 0: continue
    wc = v["SQ_WAVE_CYCLES"]
    ms = dur[k] / n / 1e6
    # VALU issue occupancy of the chip's 1024 SIMDs: quad-cycles * 4 / (GUI_ACTIVE/8 cycles * 1024)
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    busy = v["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * 1024.0) if cyc else 0.0
    print(f"{k[:36]:36s} {n:4d} {ms:7.3f} {w/n:9.0f} {v['SQ_INSTS_VALU']/w:9.0f} {v['SQ_INSTS_SALU']/w:9.0f} {v['SQ_ACTIVE_INST_VALU']/wc:8.2f} {v['SQ_ACTIVE_INST_ANY']/wc:8.2f} {v['SQ_WAIT_INST_ANY']/wc:9.2f} {v['SQ_WAIT_ANY']/wc:8.2f} {busy:14.2f}")
