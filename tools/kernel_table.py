#!/usr/bin/env python3
"""Per-kernel table of one build from the committed evidence: algorithmic bytes (bench.py's ALGO_DOUBLES), bytes moved over the fabric
(PMC passes, profiles/<tag>_traffic.json), mean launch time (rocprofv3 kernel trace, profiles/<tag>_kernel_stats.csv) and the rates they
give.  usage: tools/kernel_table.py r2_v10 2048 2048 50"""
import csv, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
tag, nx, ny, nz = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cells = nx * ny * nz; cols = nx * ny
tr = json.load(open(f"profiles/{tag}_traffic.json"))["kernels"]
st = {}
for r in csv.DictReader(open(f"profiles/{tag}_kernel_stats.csv")):
    st[r["Name"].split("(")[0].replace("void ", "")] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
bench.ALGO_DOUBLES["uv_corrector"] = bench.UV_CORRECTOR_MF
if st.get("k_vmix", (9, 0))[0] <= 1:                             # constant mixing written once: the solvers do not read the arrays (bench.py does the same)
    for g, less in bench.CONST_VMIX_LESS.items():
        bench.ALGO_DOUBLES[g] -= less
rows = []
for g, pres in bench.GROUP_KERNELS.items():
    ks = [k for k in st if any(k.startswith(p) for p in pres)]
    if not ks:
        continue
    ncall = sum(st[k][0] for k in ks)
    ms = sum(st[k][0] * st[k][1] for k in ks) / ncall             # instantiations of one launch (u / v, the modes of k2d_fused): mean over their launches
    kt = [k for k in ks if k in tr]
    moved = sum(tr[k]["launches"] * (tr[k]["read_bytes_per_launch"] + tr[k]["write_bytes_per_launch"]) for k in kt) / max(1, sum(tr[k]["launches"] for k in kt))
    algo = bench.K2D_DOUBLES * 8 * cols if g == "k2d_fused" else bench.ALGO_DOUBLES[g] * 8 * cells
    if moved < 0.2 * algo:
        continue                                                   # launches over thin strips of the box (wall and ghost faces)
    rows.append((ms * sum(st[k][0] for k in ks), g, ks[0][:28], algo, moved, ms))
print("| group | kernel | algorithmic GB | moved GB (PMC) | ms per launch | algorithmic TB/s (of 8) | moved TB/s |")
print("|---|---|---|---|---|---|---|")
for _, g, k, algo, moved, ms in sorted(rows, reverse=True):
    print(f"| {g} | `{k}` | {algo / 1e9:.2f} | {moved / 1e9:.2f} | {ms:.3f} | {algo / ms / 1e9:.2f} ({algo / ms / 1e9 / 8:.2f}) | {moved / ms / 1e9:.2f} |")
