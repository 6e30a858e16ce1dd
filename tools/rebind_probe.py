"""Step time of the bench workload on four successive allocations of the STATE BLOCK under one context (REMORA.rebind; earlier blocks
stay allocated): shows that the placement spread of DESIGN.md section 6 lives in the state block (61.4 / 60.6 / 60.3 / 59.0 ms on one box).
Run on the GPU box: python3 tools/rebind_probe.py"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import bench_case
from remora_amd.model import REMORA
case = bench_case(2048, 2048, 50, 1, 1, "DoubleGyre")
rem = REMORA(case); rem.InitData()
held = []
for t in range(4):
    if t:
        held.append(rem.rebind()); rem.InitData()
    rem.timeStep(); rem.sync()
    t0 = time.perf_counter(); rem.timeStep(); rem.timeStep(); rem.timeStep(); rem.sync()
    print("block", t, "ms/step", round((time.perf_counter() - t0) / 3 * 1e3, 2), flush=True)
