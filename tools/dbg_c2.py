"""debug: cost of the flux arithmetic in rhs_uv_3d: third-order upstream vs centred second order at the bench size"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from remora_amd.model import REMORA
for c2 in (0, 1):
    c = bench.bench_case(2048, 2048, 50, 1)
    c["uv_c2"] = c2
    r = REMORA(c); r.InitData()
    for _ in range(2): r.timeStep()
    r.timing(True); r.timing_reset()
    for _ in range(2): r.timeStep()
    rep = r.timing_report()
    print("uv_c2", c2, {k: round(v["ms"] / 2, 3) for k, v in rep.items() if k in ("rhs_uv_3d", "uv3dmix", "prestep_t3", "advance_2d")})
    r.close(); del r
    import torch; torch.cuda.empty_cache()
