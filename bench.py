#!/usr/bin/env python3
"""bench.py -- 3D baroclinic cell-updates/s of the REMORA hot path on N MI355X (one process per GPU).

A "step" is one full REMORA::Advance (setup_step + the barotropic sub-steps + advance_3d_ml, ghost exchange included)
of the DoubleGyre physics (Exec/DoubleGyre, Tests/test_files/DoubleGyre/DoubleGyre.i) on a 2048 x 2048 x 50 block per GPU
(BASELINE.json configs[3]).  With N > 1 the domain is 2048 x (2048 N) x 50, split into N j-slabs
(weak scaling; --grid PXxPY picks another i-j process grid), and every FillBoundary becomes RCCL send/recv with the neighbours.

`--workload config4` (or REMORA_BENCH_WORKLOAD=config4 in the environment, for drivers that pass no flags) runs BASELINE.json configs[4]
instead: Upwelling physics, 4096 x 2048 x 64 cells per GPU on a 4 x 2 (8 GPUs), 2 x 2 (4), 2 x 1 (2) or 1 x 1 process grid.
At N > 1 the control plane (communicator id, verdicts, timing reductions) is a gloo process group: the library's own RCCL communicator
is the only RCCL instance in the process.

Launch: `python bench.py --gpus N` from a bare shell starts its own N ranks (fresh child processes, started before this
process touches a GPU); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one rank of N.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the dominant kernel of the step (largest share of the step's time), measured live with HIP events on the
                  library's stream: algorithmic bytes per launch / mean launch duration against the 8 TB/s HBM peak
  kernels      -- the same figure for every kernel group of the step
  step_roofline-- (i) union bytes of the fused 3-D kernels / time of the 3-D part, (ii) that plus the 2-D bytes / whole step
  cpu_baseline -- the CPU oracle (oracle/, C restatement of the reference's path) timed on a bounded sample
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec

# Algorithmic doubles per cell per launch of each 3-D kernel group (distinct 3-D field components read +
# distinct 3-D field components written, each counted once; 2-D arrays, scratch, halo and cache re-reads are NOT counted).  A
# fused group counts the union of its members.
ALGO_DOUBLES = {
    "prestep_t3": 10 + 6,          # told x3, tnew x3, Hz, Huon, Hvom, W -> sstore x3, tnew x3   (3 tracers in one launch)
    "rhs_t3": 10 + 3,              # sstore x3, t x3, Huon, Hvom, Hz, W -> t x3
    "rhs_uv_3d": 8 + 2,            # uold, vold, ru, rv, Huon, Hvom, W, Hz -> ru, rv          (coriolis folded in)
    "uv3dmix": 5 + 2,              # uold, vold, Hz, u, v -> u, v
    "uv_corrector": 4 + 1,         # u, ru, Hz, Akv -> u     (AB3 + vert_visc_3d + vert_mean_3d; one launch per velocity); 5 + 2 when
                                   # update_massflux_3d rides on the sweep (+ Huon read and written), see UV_CORRECTOR_MF below
    "vert_visc_3d": 3 + 1,         # phi, Hz, Akt -> phi
    "update_massflux_3d": 3 + 1,   # phi, Hphi, Hz -> Hphi (+ phi on boundary faces only)
    "massflux_omega_new": 6 + 3,   # u, v, Huon, Hvom, Hz, z_w -> Huon, Hvom, W   (advance_3d: update_massflux_3d x2 + omega in one sweep)
    "rho_prsgrd": 4 + 3,           # T, S, Hz, z_r -> bvf, ru, rv   (k_rho_prsgrd: rho_eos + hydrostatic pressure + prsgrd in one march; rho, P stay in LDS)
    "prsgrd": 4 + 2,               # rho, P, z_r, Hz -> ru, rv
    "rho_eos": 4 + 3,              # T, S, Hz, z_r -> rho, bvf, P   (the hydrostatic pressure of prsgrd rides on the same sweep)
    "omega": 3 + 1,                # Huon, Hvom, z_w -> W
    "massflux_omega": 4 + 3,       # uold, vold, Hz, z_w -> Huon, Hvom, W   (setup_step: old mass fluxes + omega in one sweep)
    "prestep_uv": 7 + 4,           # k_prestep_uv2, u and v in one launch: uold, vold, Hz, ru x2, rv x2 -> u, v, ru, rv
    "stretch_transform": 0 + 3,
    "vmix": 0 + 4,
    "massflux_old": 3 + 2,
}
# Algorithmic doubles per COLUMN per launch of the fused barotropic kernel k2d_fused (distinct 2-D arrays, counted in
# remora_amd/csrc/rh_step2d.hip for the bench configuration: masks all one, Coriolis on).
#   every call reads   zeta(krhs), zeta(kstp), h, pm, pn, ubar(krhs), vbar(krhs), ubar(kstp), vbar(kstp), rhoS, rhoA, visc2_r,
#                      visc2_p, fcor, rufrc, rvfrc                                                                  = 16
#   predictor adds     Zt_avg1, DU_avg1, DU_avg2, DV_avg1, DV_avg2 (read + written: 5 + 5), rzeta / rubar / rvbar(krhs)
#                      written (3), zeta / ubar / vbar(knew) written (3)                                            -> 16 + 5 + 11 = 32
#   corrector adds     rzeta / rubar / rvbar(kstp) and (ptsk) read (6), DU_avg2, DV_avg2 (2 + 2), the three new levels (3)
#                                                                                                                   -> 16 + 8 + 5 = 29
# nfast predictor and nfast corrector calls run fused per step: 30.5 on average (PMC: 32 doubles per column moved).
K2D_DOUBLES = 30.5
UV_CORRECTOR_MF = 5 + 2            # u, ru, Hz, Akv, Huon -> u, Huon   (the report then has a "massflux_rest" group and no "update_massflux_3d")
# constant vertical mixing (RH_VMIX_CONST, e.g. DoubleGyre): the solvers read Akv / Akt from L1-resident rows, not from the 3-D arrays,
# and k_vmix does not run after the first step -- one read fewer per solve
CONST_VMIX_LESS = {"uv_corrector": 1, "vert_visc_3d": 1}
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic_current.json")
# kernel(s) behind each group in the rocprofv3 output (the traffic file's keys start with these)
GROUP_KERNELS = {
    "prestep_t3": ["k_tracers_tiled<0"], "rhs_t3": ["k_tracers_tiled<1"], "rhs_uv_3d": ["k_rhs_uv_tiled"], "uv3dmix": ["k_uv3dmix_tiled"],
    "uv_corrector": ["k_column_solver_reg<1, 0, 1", "k_column_solver_reg<0, 1, 1"], "vert_visc_3d": ["k_column_solver_reg<0, 0, 0"],
    "update_massflux_3d": ["k_massflux_lds"], "massflux_omega_new": ["k_massflux_omega_new"], "prsgrd": ["k_prsgrd_uv_tiled"], "rho_eos": ["k_rho_eos_P"], "rho_prsgrd": ["k_rho_prsgrd"],
    "omega": ["k_omega_lds"], "massflux_omega": ["k_massflux_omega<"], "prestep_uv": ["k_prestep_uv2"], "stretch_transform": ["k_stretch"],
    "vmix": ["k_vmix"], "massflux_old": ["k_massflux_old"], "k2d_fused": ["k2d_fused"],
}


def pmc_traffic(group, cells, build):
    """Fabric-side bytes per launch of the group's kernel from the committed PMC passes (tools/pmc_traffic.sh, collected and
    calibrated as MI355X_MICROARCH.md prescribes).  Only a file that records the SAME library build and the same grid counts:
    returns (bytes, source) or (None, reason)."""
    if not os.path.exists(TRAFFIC_FILE) or group not in GROUP_KERNELS:
        return None, "no PMC traffic file for this build"
    with open(TRAFFIC_FILE) as f:
        t = json.load(f)
    if t.get("build") != build:
        return None, f"profiles/traffic_current.json is of build {t.get('build')}, the library is {build}"
    if t.get("cells") != cells:
        return None, "profiles/traffic_current.json is of another grid size"
    tot = 0.0
    for pre in GROUP_KERNELS[group]:
        ks = [v for k, v in t["kernels"].items() if k.startswith(pre)]
        if not ks:
            return None, f"kernel {pre} not in profiles/traffic_current.json"
        # several instantiations of one kernel (u / v, the modes of k2d_fused): mean over their launches; several kernels in a group: add them
        tot += sum(v["launches"] * (v["read_bytes_per_launch"] + v["write_bytes_per_launch"]) for v in ks) / sum(v["launches"] for v in ks)
    if group == "uv_corrector":
        tot /= 2.0                  # its two entries are the two instantiations of one per-velocity launch
    return tot, f"profiles/traffic_current.json (build {build})"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nx", type=int, default=2048)
    ap.add_argument("--ny", type=int, default=2048, help="cells in j PER GPU")
    ap.add_argument("--nz", type=int, default=50)
    ap.add_argument("--case", type=str, default="DoubleGyre", help="physics of the synthetic grid: DoubleGyre (BASELINE configs[3], the measured workload) or "
                    "Upwelling (configs[4]-shaped blocks, e.g. --case Upwelling --nx 4096 --ny 2048 --nz 64)")
    ap.add_argument("--workload", type=str, default=os.environ.get("REMORA_BENCH_WORKLOAD", "configs3"), choices=["configs3", "config4"],
                    help="configs3 (default): DoubleGyre physics, --nx x --ny x --nz per GPU, 1 x N grid; config4: BASELINE configs[4], Upwelling physics "
                    "4096 x 2048 x 64 per GPU on a 4x2 / 2x2 / 2x1 / 1x1 grid (overrides --case / --nx / --ny / --nz / --grid unless --grid is given)")
    ap.add_argument("--grid", type=str, default="", help="process grid PXxPY (PX * PY = --gpus); default 1xN = j-slabs, every rank keeps full 512-B-coalesced i rows")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--place-tries", type=int, default=4, help="allocate the state up to this many times before the timed run and keep the "
                    "allocation whose probe steps ran fastest (the step time depends on which physical pages a 57 GB block lands on, +-3 %% from "
                    "process to process on this hardware); 1 = take the first")
    ap.add_argument("--cpu-sample", type=str, default="512x512x50x4", help="nx x ny x nz x steps of the CPU-oracle sample")
    ap.add_argument("--selfcheck-child", action="store_true", help="internal: one rank of the exchange-mode check that every rank of an N > 1 run "
                    "starts as a child process before it touches its GPU (see exchange_selfcheck)")
    a = ap.parse_args()
    if a.workload == "config4":
        a.case = "Upwelling"
        if (a.nx, a.ny, a.nz) == (2048, 2048, 50):          # sizes left at their defaults: the configs[4] block
            a.nx, a.ny, a.nz = 4096, 2048, 64
        if not a.grid:
            a.grid = {1: "1x1", 2: "2x1", 4: "2x2", 8: "4x2"}.get(a.gpus, f"1x{a.gpus}")
    return a


def bench_case(nx_per_gpu, ny_per_gpu, nz, px, py=None, name="DoubleGyre"):
    from remora_amd import cases
    if py is None:
        px, py = 1, px
    c = cases.get(name)
    if name in ("Upwelling", "Upwelling_GLS"):   # dx = dy = 1 km as in the reference input (its bathymetry depends on the j index and on dx)
        c["prob_hi"] = (1000.0 * nx_per_gpu * px, 1000.0 * ny_per_gpu * py, 0.0)
        c["ncell"] = (nx_per_gpu * px, ny_per_gpu * py, nz)
        return c
    base = c["ncell"]
    nx, ny = nx_per_gpu * px, ny_per_gpu * py
    # the physical extent grows with the process grid so dx, dy (and the work per GPU) are fixed
    c["prob_hi"] = (c["prob_hi"][0] * px, c["prob_hi"][1] * py, c["prob_hi"][2])
    c["ncell"] = (nx, ny, nz)
    c["fixed_dt"] = float(c["fixed_dt"] * min(base[0] / nx_per_gpu, base[1] / ny_per_gpu))    # CFL scaling
    return c


def cpu_baseline(sample):
    """Time the CPU oracle on a smaller DoubleGyre grid (same physics, same step) on the host cores of this box: the leaf passes
    (oracle/leaves.c) run OpenMP-threaded over j, ghost fills and glue serially, as in the reference, whose TimeIntegration loops
    carry no OpenMP either.  Timing build of the oracle (oracle/Makefile: liboracle_fast.so, -O3
    -march=x86-64-v3), not the -O2 -ffp-contract=off parity build."""
    from oracle import oracle_py
    oracle_py.use_fast_build()
    nx, ny, nz, steps = (int(x) for x in sample.split("x"))
    threads = max(1, min(16, int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)))   # the box's CPU share is 16
    c = bench_case(nx, ny, nz, 1)
    o = oracle_py.Oracle(c, threads=threads)
    o.advance(1)                      # first step (iic = 0 branches) not timed
    t0 = time.perf_counter()
    o.advance(steps)
    dt = time.perf_counter() - t0
    o.close()
    return {"value": nx * ny * nz * steps / dt, "unit": "cell-updates/s", "cores": threads, "kind": "port",
            "sample": f"DoubleGyre physics {nx}x{ny}x{nz} = {nx * ny * nz} cells, {steps} steps after 1 untimed ({dt:.1f} s of CPU work), oracle/ "
                      f"(C restatement of the reference's passes, gcc -O3 -march=x86-64-v3, OpenMP over the leaf passes, {threads} threads)"}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` from a bare shell: start N fresh rank processes (this parent never touches a GPU and never
    imports torch), relay rank 0's JSON line, exit with the first non-zero code.  All children are polled: when one ends badly the
    others (exactly the processes started here) are ended too instead of waiting in a rendezvous or a collective for the process
    group's timeout; stderr of every rank is inherited, so their diagnostics stay visible."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True, start_new_session=True))
    import signal
    import threading

    def end_group(p, sig):
        """the rank and whatever it started (its exchange-mode check runs in a child of its own): the process group created above"""
        try:
            os.killpg(p.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass

    out0 = []
    rd = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)     # drain rank 0's pipe while polling
    rd.start()
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            time.sleep(2.0)                                   # let the failing rank's siblings print what they have to say
            for p in procs:
                end_group(p, signal.SIGTERM)
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    end_group(p, signal.SIGKILL)              # the exact process groups started here
                    p.wait()
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    rd.join(timeout=10)
    sys.stdout.write("".join(x or "" for x in out0))
    sys.stdout.flush()
    sys.exit(rc)


SELFCHECK_TIMEOUT_S = int(os.environ.get("REMORA_BENCH_SELFCHECK_TIMEOUT", "420"))


def exchange_selfcheck(world):
    """N > 1: which exchange mode the timed run uses is decided on the node it runs on, by a CHILD process per rank started before
    this rank touches its GPU: the children form their own process group (MASTER_PORT moved aside), run two steps from the same
    initial state with the exchanges on the compute stream and two with the side-stream mode (fills of setup_step on the side
    stream; barotropic kernel split into interior and rim tiles with the exchanges under the interior), and print whether every
    rank found the fields bit-identical.  A child that fails, or that has not answered after SELFCHECK_TIMEOUT_S (a hang in a mode
    this node does not support must not take the measured run with it), is ended -- the exact process started here -- and the
    answer is "single stream".  Returns (use_side_stream, reason)."""
    port = int(os.environ.get("MASTER_PORT", "29500"))
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env["MASTER_PORT"] = str(port + 101 if port + 101 < 65000 else port - 101)
    cmd = [sys.executable, os.path.abspath(__file__), "--selfcheck-child"] + [x for x in sys.argv[1:] if x != "--selfcheck-child"]
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, timeout=SELFCHECK_TIMEOUT_S)   # ends the child on timeout
    except subprocess.TimeoutExpired:
        return False, f"the check did not finish within {SELFCHECK_TIMEOUT_S} s"
    if r.returncode != 0:
        return False, f"the check ended with exit code {r.returncode}"
    if "SELFCHECK_RESULT 1" in r.stdout:
        return True, "bit-identical to the single-stream run on this node"
    return False, "the side-stream run did NOT reproduce the single-stream one on this node"


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a.gpus)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE = {world}")
    if os.environ.get("REMORA_BENCH_TEST_DIE_RANK", "") == str(rank) and not a.selfcheck_child:      # test hook: a rank that ends before the rendezvous
        sys.exit(7)
    host_transport = os.environ.get("REMORA_BENCH_TRANSPORT", "") == "host"
    # the variable that switches the side-stream exchanges: RH_ASYNC_RCCL=1 turns them on over RCCL (off by default), RH_NO_ASYNC=1
    # turns them off over the host-staged transport of the rehearsal (on by default)
    mode_var, mode_on, mode_off = ("RH_NO_ASYNC", "0", "1") if host_transport else ("RH_ASYNC_RCCL", "1", "0")
    side_verdict = None
    wall = {}
    t_start = time.perf_counter()
    if world > 1 and not a.selfcheck_child and mode_var not in os.environ:
        side_verdict = exchange_selfcheck(world)          # before this process touches a GPU
        wall["selfcheck_s"] = round(time.perf_counter() - t_start, 2)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # REMORA_BENCH_TRANSPORT=host: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share cuda:0, messages are
    # staged through the host over gloo).  Never the measured configuration: the driver's runs use RCCL, one GPU per rank.
    if host_transport:
        local = 0
    elif world > 1 and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} needs {world} visible GPUs (found {torch.cuda.device_count()}); REMORA_BENCH_TRANSPORT=host rehearses on fewer")
    torch.cuda.set_device(local)
    if world > 1:
        # control plane (communicator id, verdicts, timing reductions: a few hundred bytes): gloo, so that the library's own RCCL
        # communicator (rh_comm_init, dlopen'd librccl) is the only RCCL instance in the process; tests/decomp_worker.py does the same
        dist.init_process_group("gloo")

    from remora_amd.model import REMORA
    px, py = (int(x) for x in a.grid.split("x")) if a.grid else (1, world)
    if px * py != world:
        raise SystemExit(f"--grid {a.grid} does not match {world} ranks")
    rx, ry = rank % px, rank // px
    case = bench_case(a.nx, a.ny, a.nz, px, py, a.case)
    nx, ny, nz = case["ncell"]
    box = ((rx * a.nx, ry * a.ny, 0), ((rx + 1) * a.nx - 1, (ry + 1) * a.ny - 1, nz - 1))
    def make():
        r = REMORA(case, device=f"cuda:{local}", box=box, nranks=(px, py), rank_xy=(rx, ry))
        if world > 1:
            if host_transport:
                r.comm_set_host_transport(dist)
            else:
                r.comm_init(dist)
        r.InitData()
        return r

    if a.selfcheck_child:
        # one rank of the check (see exchange_selfcheck): two steps each way from the same initial state, every field compared
        os.environ[mode_var] = mode_off
        ref = make(); ref.timeStep(); ref.timeStep(); ref.sync()
        os.environ[mode_var] = mode_on
        if os.environ.get("REMORA_BENCH_SELFCHECK_HANG", "") == "1":     # test hook: a side-stream mode that never comes back
            time.sleep(10 ** 6)
        rem = make(); rem.timeStep(); rem.timeStep(); rem.sync()
        same = all(torch.equal(ref.view(f)[0], rem.view(f)[0]) for f in ("cons_new", "xvel_new", "yvel_new", "ubar", "vbar", "zeta", "Zt_avg1"))
        t = torch.tensor([1 if same else 0], dtype=torch.int32, device="cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        print(f"SELFCHECK_RESULT {int(t.item())}", flush=True)
        ref.close(); rem.close()
        dist.destroy_process_group()
        return
    exchange_mode = None
    if side_verdict is not None:
        # every rank must time the same mode: the ranks' verdicts are reduced once more (a child may have failed on one rank only)
        t = torch.tensor([1 if side_verdict[0] else 0], dtype=torch.int32, device="cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        use_side = int(t.item()) == 1
        os.environ[mode_var] = mode_on if use_side else mode_off
        exchange_mode = (f"side stream: exchanges under the interior tiles ({side_verdict[1]}, checked in child processes before timing)" if use_side else
                         f"single stream ({side_verdict[1] if not side_verdict[0] else 'another rank could not confirm the side-stream mode'}: reported, not used)")
    elif world > 1:
        exchange_mode = f"{mode_var}={os.environ.get(mode_var)} (set by the caller)"
    t_make = time.perf_counter()
    rem = make()
    wall["create_and_init_s"] = round(time.perf_counter() - t_make, 2)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(r, warmup, steps):
        """W untimed steps, then K timed steps between barriers; seconds, max over ranks"""
        for _ in range(warmup):
            r.timeStep()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            r.timeStep()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    # The state as it was first allocated: what a caller that binds its arrays once (REMORA's MultiFabs through the glue) gets.  Timed
    # with the same W / K as the headline, before any placement search; reported next to it.
    first_alloc = None
    if a.place_tries > 1:
        t_fa = time.perf_counter()
        el_fa = timed(rem, a.warmup, a.steps)
        first_alloc = {"ms_per_step": el_fa / a.steps * 1e3, "value": a.nx * a.ny * a.nz * world * a.steps / el_fa}
        wall["first_allocation_run_s"] = round(time.perf_counter() - t_fa, 2)
        rem.InitData(); rem.istep = 0; rem.t_new = 0.0; rem.t_old = -1.0e200
    t_place = time.perf_counter()
    # Placement.  On this hardware the same binary on the same box runs the step 2-5 % faster or slower from one ALLOCATION of the
    # state to the next (same virtual address or not; every launch of a process alike: tools/placement_probe.py) --
    # which physical pages the block lands on decides how the ~13 concurrent streams of a kernel spread over the HBM channels.  A run
    # that cares allocates a few times at start-up and keeps the best block; so does the bench, outside the timed region, and says so.
    placement = None
    if world == 1 and a.place_tries > 1:
        def probe(r):
            r.timeStep(); r.sync()
            t0 = time.perf_counter(); r.timeStep(); r.timeStep(); r.sync()
            return (time.perf_counter() - t0) / 2 * 1e3
        need = rem.buf.numel() * 8 * 1.25
        log = [probe(rem)]
        held = []
        for _ in range(a.place_tries - 1):
            if torch.cuda.mem_get_info()[0] < need:
                break
            cand = make()
            ms = probe(cand)
            log.append(ms)
            if ms < min(log[:-1]):
                held.append(rem); rem = cand          # rejected blocks stay allocated until the search ends: the next one must land elsewhere
            else:
                held.append(cand)
        for h in held:
            h.close()
        del held
        torch.cuda.empty_cache()
        rem.InitData(); rem.istep = 0; rem.t_new = 0.0; rem.t_old = -1.0e200
        placement = {"tries": len(log), "probe_ms_per_step": [round(x, 2) for x in log], "kept": round(min(log), 2)}

    if world > 1 and a.place_tries > 1:
        # N > 1: the same search per rank, on the state block alone (the context and its communicator stay: REMORA.rebind).  The ranks
        # step together, so a rank judges a block by its own kernels only -- the event-timed groups that neither exchange nor wait
        # (HIP events on the library's stream) -- and keeps the block on which those ran fastest; every rank tries equally often.
        pure = ("prestep_t3", "rhs_t3", "rhs_uv_3d", "uv_corrector", "vert_visc_3d", "prestep_uv", "uv3dmix", "rho_eos", "prsgrd", "massflux_omega")
        red_dev = "cpu"

        def local_ms(r):
            r.timeStep(); r.sync()
            r.timing(True); r.timing_reset()
            r.timeStep(); r.timeStep()
            rep = r.timing_report(); r.timing(False)
            return sum(rep[g]["ms"] for g in pure if g in rep) / 2
        need = rem.buf.numel() * 8 * 1.25
        log = [local_ms(rem)]
        best, best_ms, held = rem.buf, log[0], []
        for _ in range(a.place_tries - 1):
            new = None
            if torch.cuda.mem_get_info()[0] >= need:
                try:
                    new = torch.zeros(rem.buf.numel(), dtype=torch.float64, device=rem.buf.device)
                except torch.OutOfMemoryError:
                    new = None
            t = torch.tensor([1 if new is not None else 0], dtype=torch.int32, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)          # one more block only if every rank got one
            if int(t.item()) == 0:
                del new
                torch.cuda.empty_cache()
                break
            held.append(rem.rebind(new))                      # earlier blocks stay allocated: the next one must land elsewhere
            rem.InitData()
            ms = local_ms(rem)
            log.append(ms)
            if ms < best_ms:
                best, best_ms = rem.buf, ms
        if rem.buf is not best:
            held.append(rem.rebind(best))
        held = [h for h in held if h is not best]
        del held
        torch.cuda.empty_cache()
        rem.InitData()
        placement = {"tries": len(log), "rank0_local_kernel_ms_per_step": [round(x, 2) for x in log], "kept": round(best_ms, 2),
                     "note": "per rank: state block re-allocated, judged by the rank's own non-exchanging kernel groups"}

    if placement is not None:
        wall["placement_search_s"] = round(time.perf_counter() - t_place, 2)
    t_timed = time.perf_counter()
    el = timed(rem, a.warmup, a.steps)
    wall["timed_run_s"] = round(time.perf_counter() - t_timed, 2)

    # sanity: the fields must still be finite (an invalid run must not produce a number)
    chk = rem.view("cons_new")[0]
    finite = bool(torch.isfinite(chk).all().item())

    # per-kernel-group timing (HIP events on the library's stream), 2 further steps
    rem.timing(True)
    rem.timing_reset()
    nprof = 2
    for _ in range(nprof):
        rem.timeStep()
    rep = rem.timing_report()
    rem.timing(False)

    cells_per_gpu = a.nx * a.ny * nz
    cols_per_gpu = a.nx * a.ny
    build = rem.build_id()
    step_ms = el / a.steps * 1e3

    def algo_bytes(g):
        if g == "k2d_fused":
            return K2D_DOUBLES * 8 * cols_per_gpu
        return ALGO_DOUBLES[g] * 8 * cells_per_gpu

    if "massflux_rest" in rep:                             # the velocity corrector carried update_massflux_3d (valid non-wall faces)
        ALGO_DOUBLES["uv_corrector"] = UV_CORRECTOR_MF
    from remora_amd import lib as L_
    if rem.params.vmix_mode == L_.VMIX_CONST and os.environ.get("RH_VMIX_REWRITE", "0") in ("", "0"):
        for g, less in CONST_VMIX_LESS.items():
            ALGO_DOUBLES[g] -= less
    # k2d_fused: one CALL of the barotropic kernel is two launches (interior, rim) when the exchange runs under it; price per call
    k2d_calls = 2 * rem.nfast * nprof
    if "k2d_fused" in rep and rep["k2d_fused"]["calls"] > k2d_calls:
        rep["k2d_fused"] = dict(rep["k2d_fused"], calls=k2d_calls)
    kern = {}
    for g, v in rep.items():
        if (g in ALGO_DOUBLES or g == "k2d_fused") and v["calls"] > 0:
            avg_ms = v["ms"] / v["calls"]
            ach = algo_bytes(g) / (avg_ms * 1e-3) / 1e9
            kern[g] = {"launches_per_step": v["calls"] / nprof, "avg_launch_ms": round(avg_ms, 4), "ms_per_step": round(v["ms"] / nprof, 4),
                       "algorithmic_GBs": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4)}
    roof = None
    if kern:
        dom = max(kern, key=lambda k: kern[k]["ms_per_step"])
        avg_ms = rep[dom]["ms"] / rep[dom]["calls"]
        ach = algo_bytes(dom) / (avg_ms * 1e-3) / 1e9
        traffic, tsrc = pmc_traffic(dom, cells_per_gpu, build)
        roof = {"kernel": GROUP_KERNELS.get(dom, [dom])[0], "group": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc, "avg_launch_ms": avg_ms,
                "algorithmic_bytes_per_launch": algo_bytes(dom), "launches_per_step": rep[dom]["calls"] / nprof,
                "share_of_step": rep[dom]["ms"] / nprof / step_ms}
    # step-level figures.  (i) 3-D part: union bytes of the fused 3-D kernels (each group's algorithmic bytes x its launches) over
    # the time of setup_step + advance_3d_ml (ghost fills, small 2-D kernels and launch gaps included).  (ii) whole step: those bytes
    # plus the barotropic kernels' bytes over the whole measured step.
    b3 = sum(ALGO_DOUBLES[g] * 8 * cells_per_gpu * rep[g]["calls"] / nprof for g in rep if g in ALGO_DOUBLES)
    b2 = K2D_DOUBLES * 8 * cols_per_gpu * rep["k2d_fused"]["calls"] / nprof if "k2d_fused" in rep else 0.0
    t3 = (rep.get("setup_step", {"ms": 0})["ms"] + rep.get("advance_3d_ml", {"ms": 0})["ms"]) / nprof
    t_all = sum(rep.get(k, {"ms": 0})["ms"] for k in ("setup_step", "advance_2d", "advance_3d_ml")) / nprof
    step_roof = {"bytes_3d_union": b3, "doubles_per_cell_3d_union": b3 / 8 / cells_per_gpu, "ms_3d": round(t3, 3),
                 "frac_3d": (b3 / (t3 * 1e-3) / 1e9 / HBM_PEAK_GBS) if t3 > 0 else None,
                 "bytes_2d": b2, "ms_step_profiled": round(t_all, 3),
                 "frac_step": ((b3 + b2) / (t_all * 1e-3) / 1e9 / HBM_PEAK_GBS) if t_all > 0 else None,
                 "frac_3d_at_round2_bytes": (114.0 * 8 * cells_per_gpu / (t3 * 1e-3) / 1e9 / HBM_PEAK_GBS) if t3 > 0 else None,
                 "frac_3d_reference_granular": (1232.0 * cells_per_gpu / (t3 * 1e-3) / 1e9 / HBM_PEAK_GBS) if t3 > 0 else None,
                 "note": "frac_3d = union bytes of the fused 3-D kernels / (setup_step + advance_3d_ml time); frac_3d_at_round2_bytes prices the same time "
                         "against the 114 doubles per cell the round-2 kernels moved algorithmically (this round removed bytes: rho / P never stored, "
                         "constant mixing never re-read), for comparison across rounds; frac_step adds the k2d_fused bytes "
                         "and divides by the whole step; frac_3d_reference_granular prices the same time against the 1232 B/cell the "
                         "reference's own pass structure moves (SURVEY.md 8d)"}
    # the barotropic loop: its share of the step, launches per call and what a call costs outside the fused kernel (launch gaps,
    # the first and last calls' small kernels, ghost fills of open or periodic sides)
    baro = None
    if "k2d_fused" in rep and "advance_2d" in rep:
        n2 = rep["advance_2d"]["calls"] / nprof
        baro = {"ms_per_step": round(rep["advance_2d"]["ms"] / nprof, 3), "k2d_fused_ms_per_step": round(rep["k2d_fused"]["ms"] / nprof, 3),
                "calls_per_step": n2, "overhead_us_per_call": round((rep["advance_2d"]["ms"] - rep["k2d_fused"]["ms"]) / nprof / n2 * 1e3, 2)}
    if rank == 0:
        out = {"metric": "3D baroclinic cell-updates/s", "value": cells_per_gpu * world * a.steps / el, "unit": "cell-updates/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": step_ms, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"{'BASELINE configs[4]: ' if a.workload == 'config4' else ''}{a.case} physics, {a.nx}x{a.ny}x{nz} cells per GPU (domain {nx}x{ny}x{nz}), "
                                      f"dt {case['fixed_dt']:.3f} s, ndtfast {case['ndtfast']}, nfast {rem.nfast}, "
                                      f"{f'{px}x{py} i-j decomposition, ghost exchange over ' + ('gloo (host-staged)' if host_transport else 'RCCL') if world > 1 else 'single box'}",
                          "parallelism": f"ij-decomp {px}x{py}"},
               **({"ms_per_step_first_allocation": first_alloc["ms_per_step"], "value_first_allocation": first_alloc["value"],
                   "first_allocation_note": "same W / K on the state block as first allocated (what a caller that binds its arrays once sees); "
                                            "`value` is timed on the block the placement search kept"} if first_alloc else {}),
               "wall_s": dict(wall, total_s=round(time.perf_counter() - t_start, 2)),
               "finite": finite, "build": build, **({"exchange": exchange_mode} if exchange_mode else {}), **({"placement": placement} if placement else {}), **({"transport": "host-staged gloo rehearsal (NOT the RCCL configuration)"} if host_transport else {}),
               "roofline": roof, "step_roofline": step_roof, **({"barotropic_loop": baro} if baro else {}), "kernels": kern,
               "kernel_ms_per_step": {k: round(v["ms"] / nprof, 4) for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])}}
        if not a.no_cpu_baseline and world == 1:          # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(a.cpu_sample)
        if not finite:
            out["value"] = 0.0
            out["error"] = "non-finite field after the timed steps"
        print(json.dumps(out), flush=True)
    rem.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
