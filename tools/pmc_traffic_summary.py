#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from the two PMC passes of tools/pmc_traffic.sh.

FETCH_SIZE / WRITE_SIZE are uncalibrated for 8-B-per-lane accesses (MI355X_MICROARCH.md, HBM section), so both are calibrated on
kernels of this very run whose byte counts are known exactly and which use the same access width: k_copy (reads n*8 B, writes
n*8 B; the eight old/new state copies of rh_init_state and of the first setup_step).  usage: pmc_traffic_summary.py TAG nx ny nz out.json"""
import csv, collections, glob, json, sys

tag, nx, ny, nz, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]


def agg(counter):
    path = glob.glob(f"gpurun_out/pmc_{tag}_{counter}/*counter_collection.csv")[0]
    val = collections.defaultdict(float); cnt = collections.defaultdict(int); dur = collections.defaultdict(float)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        val[k] += float(row["Counter_Value"]); cnt[k] += 1
        dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    return val, cnt, dur


fs, fc, fd = agg("FETCH_SIZE")
ws, wc, wd = agg("WRITE_SIZE")
# known bytes of the 8 k_copy launches: cons (3 comps, ng 3), xvel, yvel, zvel, each copied twice
g = 6
n_copy = 2 * ((nx + g) * (ny + g) * nz * 3 + (nx + 1 + g) * (ny + g) * nz + (nx + g) * (ny + 1 + g) * nz + (nx + g) * (ny + g) * (nz + 1))
known = n_copy * 8.0
assert fc["k_copy"] == 8 and wc["k_copy"] == 8, (fc["k_copy"], wc["k_copy"])
cal_r = known / fs["k_copy"]; cal_w = known / ws["k_copy"]
builds = set()
for c in ("FETCH_SIZE", "WRITE_SIZE"):      # the library build both passes ran (bench.py prints it in its JSON line)
    for line in open(f"gpurun_out/pmc_{tag}_{c}/bench.json"):
        if line.startswith("{"):
            builds.add(json.loads(line).get("build"))
assert len(builds) == 1, builds
res = {"build": builds.pop(), "calibration": {"kernel": "k_copy x8", "known_bytes_each_way": known, "bytes_per_FETCH_SIZE_unit": cal_r, "bytes_per_WRITE_SIZE_unit": cal_w},
       "cells": nx * ny * nz, "kernels": {}}
cells = nx * ny * nz
print(f"calibration: 1 FETCH_SIZE unit = {cal_r:.1f} B, 1 WRITE_SIZE unit = {cal_w:.1f} B")
print(f"{'kernel':34s} {'n':>5s} {'rd B/cell':>10s} {'wr B/cell':>10s} {'rd+wr GB/launch':>16s}")
for k in sorted(fs, key=lambda k: -fd[k]):
    if not k.startswith("k"):
        continue
    n = fc[k]
    rd = fs[k] * cal_r / n; wr = ws.get(k, 0.0) * cal_w / max(wc.get(k, 1), 1)
    res["kernels"][k] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "pmc_pass_ms_per_launch": fd[k] / n / 1e6}
    print(f"{k[:34]:34s} {n:5d} {rd / cells:10.1f} {wr / cells:10.1f} {(rd + wr) / 1e9:16.3f}")
json.dump(res, open(out, "w"), indent=1)
