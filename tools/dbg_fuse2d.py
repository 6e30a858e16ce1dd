"""debug: fused vs three-kernel barotropic call, field by field (run on the GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remora_amd import cases
from remora_amd.model import REMORA
case = sys.argv[1] if len(sys.argv) > 1 else "Upwelling"
ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = cases.get(case)
os.environ["RH_NOFUSE2D"] = "1"
a = REMORA(cfg); a.InitData()
os.environ["RH_NOFUSE2D"] = "0"
b = REMORA(cfg); b.InitData()
F = ["zeta", "ubar", "vbar", "Zt_avg1", "DU_avg1", "DU_avg2", "DV_avg1", "DV_avg2", "rubar", "rvbar", "rzeta", "ru2d", "rv2d", "rufrc", "rvfrc"]
for r in (a, b):
    r.begin_step(); r.setup_step()
for q in range(ncalls):
    for r in (a, b):
        r.advance_2d_onestep(q)
    bad = False
    for f in F:
        x, lo, hi = a.get(f); y, _, _ = b.get(f)
        if not np.array_equal(x, y):
            bad = True
            w = np.argwhere(x != y)
            print(f"call {q} field {f} lo {lo} hi {hi}: {len(w)} diffs, max {np.nanmax(np.abs(x - y)):.3e}; comps {sorted(set(w[:,0]))} "
                  f"i {sorted(set(w[:,3] + lo[0]))[:12]} j {sorted(set(w[:,2] + lo[1]))[:12]}")
    if bad:
        break
print("done", case)
