"""Reads a rocprofv3 kernel trace (CSV) and reports how much of the ghost-fill kernels' time (k_fill_periodic, k_xblocks, k_bc_pass,
k_wrap_rows) ran concurrently with another kernel of the step: evidence that the side-stream fills overlap compute."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows)
fill = ("k_fill_periodic", "k_xblocks", "k_bc_pass", "k_wrap_rows")
tot = 0; ov = 0; n = 0; novl = 0
streams = {}
for q, (s, e, name, st) in enumerate(iv):
    if not name.startswith(fill):
        continue
    streams[st] = streams.get(st, 0) + 1
    n += 1; tot += e - s
    o = 0
    for (s2, e2, n2, st2) in iv[max(0, q - 8):q + 8]:
        if (s2, e2, n2) == (s, e, name) or n2.startswith(fill):
            continue
        o = max(o, min(e, e2) - max(s, s2))
    if o > 0:
        ov += o; novl += 1
print(f"ghost-fill kernels: {n} launches, {tot / 1e6:.3f} ms in total, by stream/queue {streams}")
print(f"overlapping a compute kernel: {novl} launches, {ov / 1e6:.3f} ms ({100.0 * ov / max(tot, 1):.1f} % of their time)")
