"""bench.py with more than one rank, rehearsed on the one-GPU box (REMORA_BENCH_TRANSPORT=host: the ranks share cuda:0 and the ghost
exchange is staged through the host over gloo; the driver's runs use RCCL, one GPU per rank).  What is checked is the bench's own
machinery around the timed steps: the self-spawned ranks, the exchange-mode check that every rank runs in a child process before it
touches the GPU, its watchdog, and the JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, timeout=420, torchrun=False):
    env = dict(os.environ, REMORA_BENCH_TRANSPORT="host", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", **extra_env)
    for k in ("RH_NO_ASYNC", "RH_ASYNC_RCCL", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    head = [sys.executable]
    if torchrun:                    # the driver's launch line for N > 1
        head += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29741"]
    cmd = head + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--nx", "256", "--ny", "160", "--nz", "6", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--place-tries", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    return json.loads(lines[0])


@pytest.mark.parametrize("torchrun", [False, True])
def test_two_ranks_pick_the_side_stream_mode_after_checking_it(torchrun):
    d = _bench({}, torchrun=torchrun)
    assert d["n_gpus"] == 2 and d["finite"] and d["value"] > 0 and d["scaling"] == "weak"
    assert d["exchange"].startswith("side stream"), d["exchange"]
    assert d["placement"]["tries"] == 2 and d["placement"]["kept"] > 0          # per-rank search over state blocks (REMORA.rebind)
    # the barotropic kernel ran split into interior and rim tiles: two launches per fused call, priced as one call
    import re
    nfast = int(re.search(r"nfast (\d+)", d["config"]["workload"]).group(1))
    assert d["kernels"]["k2d_fused"]["launches_per_step"] == 2 * nfast


def test_a_check_that_never_answers_falls_back_to_the_single_stream():
    d = _bench({"REMORA_BENCH_SELFCHECK_HANG": "1", "REMORA_BENCH_SELFCHECK_TIMEOUT": "25"})
    assert d["n_gpus"] == 2 and d["finite"] and d["value"] > 0
    assert d["exchange"].startswith("single stream (the check did not finish"), d["exchange"]
