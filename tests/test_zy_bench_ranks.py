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


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(extra_env, timeout=420, torchrun=False, args=None, expect_rc=0):
    env = dict(os.environ, REMORA_BENCH_TRANSPORT="host", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", **extra_env)
    for k in ("RH_NO_ASYNC", "RH_ASYNC_RCCL", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    head = [sys.executable]
    if torchrun:                    # the driver's launch line for N > 1
        head += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
    cmd = head + [os.path.join(ROOT, "bench.py"), "--gpus", "2"] + (args or ["--nx", "256", "--ny", "160", "--nz", "6"]) + ["--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--place-tries", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if expect_rc:
        assert r.returncode == expect_rc and not lines, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
        return None
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    return json.loads(lines[0])


@pytest.mark.parametrize("torchrun", [False, True])
def test_two_ranks_pick_the_side_stream_mode_after_checking_it(torchrun):
    d = _bench({}, torchrun=torchrun)
    assert d["n_gpus"] == 2 and d["finite"] and d["value"] > 0 and d["scaling"] == "weak"
    assert d["exchange"].startswith("side stream"), d["exchange"]
    assert d["placement"]["tries"] == 2 and d["placement"]["kept"] > 0          # per-rank search over state blocks (REMORA.rebind)
    # the figure of the first allocation is reported next to the searched one, and the wall time of every phase around the timed run
    assert d["ms_per_step_first_allocation"] > 0 and d["value_first_allocation"] > 0
    assert {"selfcheck_s", "create_and_init_s", "placement_search_s", "timed_run_s", "total_s"} <= set(d["wall_s"])
    # the barotropic kernel ran split into interior and rim tiles: two launches per fused call, priced as one call
    import re
    nfast = int(re.search(r"nfast (\d+)", d["config"]["workload"]).group(1))
    assert d["kernels"]["k2d_fused"]["launches_per_step"] == 2 * nfast


def test_a_check_that_never_answers_falls_back_to_the_single_stream():
    d = _bench({"REMORA_BENCH_SELFCHECK_HANG": "1", "REMORA_BENCH_SELFCHECK_TIMEOUT": "25"})
    assert d["n_gpus"] == 2 and d["finite"] and d["value"] > 0
    assert d["exchange"].startswith("single stream (the check did not finish"), d["exchange"]


def test_config4_workload_is_reachable_without_size_flags():
    """BASELINE configs[4] (Upwelling physics, 4096 x 2048 x 64 per GPU, 2 x 1 grid at two ranks) selected by the environment, as a
    driver that passes only --gpus would; here at a rehearsal size over the host transport"""
    d = _bench({"REMORA_BENCH_WORKLOAD": "config4"}, args=["--nx", "192", "--ny", "128", "--nz", "8"])
    assert d["n_gpus"] == 2 and d["finite"] and d["value"] > 0
    w = d["config"]["workload"]
    assert w.startswith("BASELINE configs[4]: Upwelling physics, 192x128x8 cells per GPU (domain 384x128x8)") and "2x1 i-j decomposition" in w, w
    assert d["config"]["parallelism"] == "ij-decomp 2x1"


def test_a_rank_that_dies_ends_the_run():
    """bench.py --gpus 2 from a bare shell: rank 1 ends before the rendezvous; the parent must end rank 0 and return the failure
    instead of leaving rank 0 in the rendezvous until the process group's timeout"""
    import time
    t0 = time.time()
    _bench({"REMORA_BENCH_TEST_DIE_RANK": "1"}, timeout=120, expect_rc=7)
    assert time.time() - t0 < 100
