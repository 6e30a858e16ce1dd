"""Does the step time depend on where the state block lands?  One process, several allocations of the bench state in a row (fresh
hipMalloc each time), a few timed steps on each."""
import sys, time
sys.path.insert(0, ".")
import torch
from bench import bench_case
from remora_amd.model import REMORA

case = bench_case(2048, 2048, 50, 1, 1, "DoubleGyre")
hold = []
for trial in range(6):
    rem = REMORA(case, device="cuda:0")
    rem.InitData()
    for _ in range(2):
        rem.timeStep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        rem.timeStep()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 6 * 1e3
    print(f"trial {trial}: {ms:.2f} ms/step, state at {rem.buf.data_ptr():#x}", flush=True)
    if trial in (2, 3):            # keep these blocks alive so the next one must land elsewhere
        hold.append(rem.buf)
    rem.close()
    del rem
    torch.cuda.empty_cache()
