"""Where in the 288 GB does a fast state block live?  One process: the state allocated first (low addresses), then again behind
ballast blocks of growing size that are held while it is allocated and probed; prints ms per step and the block's address."""
import sys, time
sys.path.insert(0, ".")
import torch
from bench import bench_case
from remora_amd.model import REMORA

case = bench_case(2048, 2048, 50, 1, 1, "DoubleGyre")


def probe(tag, ballast_gb):
    ballast = torch.empty(int(ballast_gb * 2**30), dtype=torch.uint8, device="cuda:0") if ballast_gb > 0 else None
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
    print(f"{tag}: ballast {ballast_gb:5.1f} GB at {ballast.data_ptr() if ballast is not None else 0:#x}, state at {rem.buf.data_ptr():#x}: {ms:.2f} ms/step", flush=True)
    rem.close()
    del rem, ballast
    torch.cuda.empty_cache()


for rep in range(2):
    for gb in (0, 8, 32, 64, 128, 190, 0):
        probe(f"rep {rep}", gb)
