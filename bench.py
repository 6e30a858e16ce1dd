#!/usr/bin/env python3
"""bench.py -- 3D baroclinic cell-updates/s of the REMORA hot path on N MI355X (one process per GPU).

A "step" is one full REMORA::Advance (setup_step + the barotropic sub-steps + advance_3d_ml, ghost exchange included)
of the DoubleGyre physics (Exec/DoubleGyre, Tests/test_files/DoubleGyre/DoubleGyre.i) on a 2048 x 2048 x 50 block per GPU
(BASELINE.json configs[3]).  With N > 1 the domain is 2048 x (2048 N) x 50, split into N j-slabs
(weak scaling; --grid PXxPY picks another i-j process grid), and every FillBoundary becomes RCCL send/recv with the neighbours.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the dominant kernel group of the step, measured live with HIP events on the library's stream
  cpu_baseline -- the CPU oracle (oracle/, a single-threaded restatement of the reference's path) timed on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec

# Algorithmic doubles per cell per launch of each kernel group (distinct 3-D field components read + distinct
# 3-D field components written, each counted once; 2-D arrays, scratch, halo and cache re-reads are NOT counted).  A fused group
# counts the union of its members.
ALGO_DOUBLES = {
    "prestep_t3": 10 + 6,          # told x3, tnew x3, Hz, Huon, Hvom, W -> sstore x3, tnew x3   (3 tracers in one launch)
    "rhs_t3": 10 + 3,              # sstore x3, t x3, Huon, Hvom, Hz, W -> t x3
    "rhs_uv_3d": 8 + 2,            # uold, vold, ru, rv, Huon, Hvom, W, Hz -> ru, rv          (coriolis folded in)
    "uv3dmix": 5 + 2,              # uold, vold, Hz, u, v -> u, v
    "uv_corrector": 4 + 1,         # u, ru, Hz, Akv -> u     (AB3 + vert_visc_3d + vert_mean_3d; one launch per velocity)
    "vert_visc_3d": 3 + 1,         # phi, Hz, Akt -> phi
    "update_massflux_3d": 3 + 1,   # phi, Hphi, Hz -> Hphi (+ phi on boundary faces only)
    "prsgrd": 4 + 2,               # rho, P, z_r, Hz -> ru, rv
    "rho_eos": 4 + 3,              # T, S, Hz, z_r -> rho, bvf, P   (the hydrostatic pressure of prsgrd rides on the same sweep)
    "omega": 3 + 1,                # Huon, Hvom, z_w -> W
    "massflux_omega": 4 + 3,       # uold, vold, Hz, z_w -> Huon, Hvom, W   (setup_step: old mass fluxes + omega in one sweep)
    "prestep_uv": 4 + 3,           # vel_old, Hz, rvel x2 -> vel, rvel x2
    "stretch_transform": 0 + 3,
    "vmix": 0 + 4,
    "massflux_old": 3 + 2,
}
# kernel(s) behind each group in the rocprofv3 output (profiles/*_traffic.json keys start with these)
GROUP_KERNELS = {
    "prestep_t3": ["k_tracers_tiled<0"], "rhs_t3": ["k_tracers_tiled<1"], "rhs_uv_3d": ["k_rhs_uv_tiled"], "uv3dmix": ["k_uv3dmix_tiled"],
    "uv_corrector": ["k_column_solver_reg<1, 0, 1", "k_column_solver_reg<0, 1, 1"], "vert_visc_3d": ["k_column_solver_reg<0, 0, 0"],
    "update_massflux_3d": ["k_massflux_lds"], "prsgrd": ["k_prsgrd_uv_tiled"], "rho_eos": ["k_rho_eos_P"], "omega": ["k_omega_lds"], "massflux_omega": ["k_massflux_omega"],
    "prestep_uv": ["k_prestep_uv"], "stretch_transform": ["k_stretch"], "vmix": ["k_vmix"], "massflux_old": ["k_massflux_old"],
}


def pmc_traffic(group, cells):
    """HBM-side bytes per launch of the group's kernel from the committed PMC passes (profiles/r1_v10_traffic.json, collected with
    tools/pmc_traffic.sh on this same workload and calibrated as MI355X_MICROARCH.md prescribes); None if absent or another size."""
    path = os.path.join(ROOT, "profiles", "r1_v10_traffic.json")
    if not os.path.exists(path) or group not in GROUP_KERNELS:
        return None
    with open(path) as f:
        t = json.load(f)
    if t.get("cells") != cells:
        return None
    tot = 0.0; n = 0
    for pre in GROUP_KERNELS[group]:
        ks = [v for k, v in t["kernels"].items() if k.startswith(pre)]
        if not ks:
            return None
        # several instantiations of one kernel (u / v): average them; several kernels in a group (prsgrd): add them
        tot += sum(v["read_bytes_per_launch"] + v["write_bytes_per_launch"] for v in ks) / len(ks)
        n += 1
    if group == "uv_corrector":
        tot /= 2.0                  # its two entries are the two instantiations of one per-velocity launch
    return tot


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
    ap.add_argument("--grid", type=str, default="", help="process grid PXxPY (PX * PY = --gpus); default 1xN = j-slabs, every rank keeps full 512-B-coalesced i rows")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=str, default="256x256x50x3", help="nx x ny x nz x steps of the CPU-oracle sample")
    return ap.parse_args()


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
    """Time the CPU oracle on a small DoubleGyre grid (same physics, same step) on the host cores of this box: the leaf passes
    (oracle/leaves.c) run OpenMP-threaded over j, ghost fills and glue serially, as in the reference, whose TimeIntegration loops
    carry no OpenMP either.  The threaded run is bitwise the serial one (tests/test_oracle_gold.py)."""
    from oracle.oracle_py import Oracle
    nx, ny, nz, steps = (int(x) for x in sample.split("x"))
    threads = max(1, min(16, int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)))   # the box's CPU share is 16
    c = bench_case(nx, ny, nz, 1)
    o = Oracle(c, threads=threads)
    o.advance(1)                      # first step (iic = 0 branches) not timed
    t0 = time.perf_counter()
    o.advance(steps)
    dt = time.perf_counter() - t0
    o.close()
    return {"value": nx * ny * nz * steps / dt, "unit": "cell-updates/s", "cores": threads, "kind": "port",
            "sample": f"DoubleGyre physics {nx}x{ny}x{nz}, {steps} steps after 1 untimed ({dt:.1f} s), oracle/ (C restatement, -O2, "
                      f"OpenMP over the leaf passes, {threads} threads)"}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # REMORA_BENCH_TRANSPORT=host: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share cuda:0, messages are
    # staged through the host over gloo).  Never the measured configuration: the driver's runs use RCCL, one GPU per rank.
    host_transport = os.environ.get("REMORA_BENCH_TRANSPORT", "") == "host"
    if host_transport:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if host_transport:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))

    from remora_amd.model import REMORA
    px, py = (int(x) for x in a.grid.split("x")) if a.grid else (1, world)
    if px * py != world:
        raise SystemExit(f"--grid {a.grid} does not match {world} ranks")
    rx, ry = rank % px, rank // px
    case = bench_case(a.nx, a.ny, a.nz, px, py, a.case)
    nx, ny, nz = case["ncell"]
    box = ((rx * a.nx, ry * a.ny, 0), ((rx + 1) * a.nx - 1, (ry + 1) * a.ny - 1, nz - 1))
    rem = REMORA(case, device=f"cuda:{local}", box=box, nranks=(px, py), rank_xy=(rx, ry))
    if world > 1:
        if host_transport:
            rem.comm_set_host_transport(dist)
        else:
            rem.comm_init(dist)
    rem.InitData()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        rem.timeStep()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        rem.timeStep()
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cpu" if host_transport else f"cuda:{local}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

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
    roof = None
    leaf = {k: v for k, v in rep.items() if k in ALGO_DOUBLES}
    if leaf:
        dom = max(leaf, key=lambda k: leaf[k]["ms"])
        avg_ms = leaf[dom]["ms"] / leaf[dom]["calls"]
        bytes_launch = ALGO_DOUBLES[dom] * 8 * cells_per_gpu
        ach = bytes_launch / (avg_ms * 1e-3) / 1e9
        roof = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": pmc_traffic(dom, cells_per_gpu), "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": bytes_launch,
                "share_of_step": leaf[dom]["ms"] / nprof / (el / a.steps * 1e3)}
    if rank == 0:
        out = {"metric": "3D baroclinic cell-updates/s", "value": cells_per_gpu * world * a.steps / el, "unit": "cell-updates/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"{a.case} physics, {a.nx}x{a.ny}x{nz} cells per GPU (domain {nx}x{ny}x{nz}), "
                                      f"dt {case['fixed_dt']:.3f} s, ndtfast {case['ndtfast']}, nfast {rem.nfast}, "
                                      f"{f'{px}x{py} i-j decomposition, ghost exchange over RCCL' if world > 1 else 'single box'}",
                          "parallelism": f"ij-decomp {px}x{py}"},
               "finite": finite, **({"transport": "host-staged gloo rehearsal (NOT the RCCL configuration)"} if host_transport else {}),
               "roofline": roof,
               "kernel_ms_per_step": {k: round(v["ms"] / nprof, 4) for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])},
               "algorithmic_hbm_frac_step": (1232.0 * cells_per_gpu / (el / a.steps)) / 1e9 / HBM_PEAK_GBS}
        if not a.no_cpu_baseline:
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
