"""i-j decomposition on the GPU: 2 to 4 ranks (processes sharing the box's single GPU, messages through the transport hook
over gloo) on j-slab, i-slab and 2 x 2 process grids must reproduce the single-box run bit for bit in every valid region."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("case,world,port,grid", [("Seamount", 2, 29611, "1x2"), ("Upwelling", 3, 29612, "1x3"), ("DoubleGyre", 3, 29613, "1x3"),
                                                  ("Advection", 2, 29614, "1x2"), ("Seamount", 4, 29615, "2x2"), ("Upwelling", 2, 29616, "2x1"),
                                                  ("DoubleGyre", 4, 29617, "2x2"), ("Advection", 3, 29618, "3x1"),
                                                  ("Upwelling_GLS", 2, 29619, "1x2"), ("Upwelling_GLS", 4, 29620, "2x2"),
                                                  # open (radiating / outflow) sides on all four faces, GLS on: the physical-boundary passes of a
                                                  # box must act on the domain's faces only, and the radiation condition reads the previous level
                                                  ("Channel_Test_orlanski", 2, 29621, "1x2"), ("Channel_Test_orlanski", 2, 29622, "2x1"),
                                                  ("Channel_Test_orlanski", 4, 29623, "2x2")])
def test_decomposed_equals_single_box(case, world, port, grid):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "decomp_worker.py"), case, "3", grid]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "DECOMP_OK" in out, out[-4000:]


@pytest.mark.parametrize("split", ["1", "0"])
@pytest.mark.parametrize("case,world,port,grid", [("DoubleGyre@420x100x4", 2, 29651, "2x1"), ("Upwelling@300x210x6", 2, 29652, "1x2"),
                                                  ("DoubleGyre@420x210x4", 4, 29653, "2x2"), ("Seamount@300x210x5", 3, 29654, "1x3"),
                                                  ("DoubleGyre@256x256x4", 4, 29655, "1x4")])      # j-slabs with inner ranks, the bench's default grid shape
def test_interior_rim_split_keeps_the_bits(case, world, port, grid, split):
    """Boxes wide enough for k2d_fused to have interior tiles (>= 3 tiles of 63 x 7 points each way): inside rh_advance the barotropic
    loop then launches the interior tiles first, runs the flux frame, the DUon / DVom exchange and the previous call's ghost fill on
    the side stream under them, and the rim tiles last.  The worker asserts the launch count (two k2d_fused launches per fused call
    with the split, one with RH_NO_SPLIT2D=1) and the usual bit-identity with the single-box run."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", REMORA_DECOMP_EXPECT_SPLIT=split,
               RH_NO_SPLIT2D="0" if split == "1" else "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port + (10 if split == "0" else 0)), os.path.join(ROOT, "tests", "decomp_worker.py"), case, "2", grid]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "DECOMP_OK" in out, out[-4000:]


@pytest.mark.parametrize("case,world,port,grid", [("Seamount", 4, 29661, "2x2"), ("DoubleGyre", 3, 29662, "1x3"), ("Upwelling@300x210x6", 2, 29663, "1x2"),
                                                  ("Upwelling_GLS", 2, 29664, "2x1")])
def test_decomposed_with_land_equals_single_box(case, world, port, grid):
    """Land in the domain (tests/landmask.py: islands, coasts on walls and across periodic sides, cut by the box boundaries): every
    box writes its masks from the same layout, rh_update_mskp builds mskp and exchanges the mask ghost cells, and three steps must
    equal the single-box run with the same land bit for bit (the third case is wide enough for the interior / rim split)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", REMORA_DECOMP_LAND="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "decomp_worker.py"), case, "3", grid]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "DECOMP_OK" in out, out[-4000:]


@pytest.mark.parametrize("case,world,port,grid", [("DoubleGyre", 2, 29671, "1x2"), ("DoubleGyre", 2, 29672, "2x1"), ("Seamount", 4, 29673, "2x2")])
def test_update_mskp_fills_the_seams(case, world, port, grid):
    """Every box writes mskr / msku / mskv over its VALID box only (plus the ghost cells beyond a wall of the domain), as
    include/remora_hip.h asks; the ghost cells at rank seams keep init_masks' 1.0.  rh_update_mskp must bring the neighbour's cells
    in before it forms mskp on its first valid node line (the reference's FAB intersection copy, init_from_netcdf.cpp:400, does):
    the masks, mskp included, and three steps must equal the single-box run bit for bit."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", REMORA_DECOMP_LAND="valid")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "decomp_worker.py"), case, "3", grid]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "DECOMP_OK" in out, out[-4000:]


def _ngpu():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("async_rccl", ["0", "1"])
@pytest.mark.parametrize("case,world,port,grid", [("Seamount", 2, 29631, "1x2"), ("Upwelling", 2, 29632, "2x1"), ("DoubleGyre", 2, 29633, "1x2"),
                                                  ("Seamount", 4, 29634, "2x2"), ("DoubleGyre", 4, 29635, "2x2")])
def test_decomposed_equals_single_box_over_rccl(case, world, port, grid, async_rccl):
    """The same check over the library's built-in RCCL transport (ncclSend / ncclRecv, rh_comm.hip), one GPU per rank, with the
    side-stream fills off and on (RH_ASYNC_RCCL).  Needs as many GPUs as ranks: skipped on a one-GPU box."""
    if _ngpu() < world:
        pytest.skip(f"{world} GPUs needed, {_ngpu()} visible")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", REMORA_DECOMP_TRANSPORT="rccl",
               RH_ASYNC_RCCL=async_rccl)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port + (10 if async_rccl == "1" else 0)), os.path.join(ROOT, "tests", "decomp_worker.py"), case, "3", grid]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "DECOMP_OK" in out, out[-4000:]
