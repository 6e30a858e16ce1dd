"""j-slab decomposition on the GPU: 2 and 3 ranks (processes sharing the box's single GPU, messages through the transport
hook over gloo) must reproduce the single-box run bit for bit in every valid region."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("case,world,port", [("Seamount", 2, 29611), ("Upwelling", 3, 29612), ("DoubleGyre", 3, 29613), ("Advection", 2, 29614)])
def test_decomposed_equals_single_box(case, world, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "decomp_worker.py"), case, "3"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "DECOMP_OK" in out, out[-4000:]
