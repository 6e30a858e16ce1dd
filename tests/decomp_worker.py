"""Worker of tests/test_gpu_decomp.py: one rank of an i-j decomposed run (all ranks share cuda:0; messages travel through
the transport hook + gloo; argv[3] = "PXxPY" process grid, default 1 x world = j-slabs).  Rank 0 also runs the same case as a single box and checks that every rank's valid region is
BIT-IDENTICAL to it (multi-GPU must reproduce the 1-GPU result)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remora_amd import cases                     # noqa: E402
from remora_amd.model import REMORA              # noqa: E402

FIELDS = ["cons_new", "cons_old", "xvel_new", "yvel_new", "ubar", "vbar", "zeta", "Zt_avg1", "Hz", "z_w", "Huon", "Hvom", "ru", "rv",
          "DU_avg1", "DU_avg2", "DV_avg1", "DV_avg2", "rubar", "rvbar", "rzeta", "ru2d", "rv2d", "sstore", "rhoA", "rhoS", "Akv"]


def slab_rows(ny, world, rank):
    base, rem = divmod(ny, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0) - 1


def valid_part(rem, name, jlo, jhi, ny):
    """the rows [jlo, jhi (+1 for j-nodal arrays on the last slab)] of a field, all i inside the domain, all k"""
    a, lo, hi = rem.get(name)
    nx = rem.ncell[0]
    ng0 = -lo[0]
    nod_i = (hi[0] - lo[0] + 1) - 2 * ng0 - nx
    return a, lo, hi, nod_i


def main():
    case, nsteps = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # "Case@NXxNYxNZ": the case's physics on a finer grid (cases.scaled), for boxes wide enough to have interior tiles
    if "@" in case:
        case, dims = case.split("@")
        cfg = cases.scaled(case, tuple(int(x) for x in dims.split("x")))
    else:
        cfg = cases.get(case)
    nx, ny, nz = cfg["ncell"]
    px, py = (int(x) for x in sys.argv[3].split("x")) if len(sys.argv) > 3 else (1, world)
    assert px * py == world
    rx, ry = rank % px, rank // px
    jlo, jhi = slab_rows(ny, py, ry)
    ilo, ihi = slab_rows(nx, px, rx)
    land = os.environ.get("REMORA_DECOMP_LAND", "") in ("1", "valid")       # land in the domain: masks written by every box from the same layout
    land_valid_only = os.environ.get("REMORA_DECOMP_LAND", "") == "valid"   # ... over the valid boxes only: rh_update_mskp must fill the seams
    if land:
        cfg = dict(cfg, masks_all_one=0)

    # REMORA_DECOMP_TRANSPORT=rccl: the library's built-in RCCL send/recv, one GPU per rank (LOCAL_RANK); the process group stays
    # gloo (it only carries the 128-byte communicator id and the gathers of this script).  Default: all ranks on cuda:0, host-staged.
    rccl = os.environ.get("REMORA_DECOMP_TRANSPORT", "") == "rccl"
    dev = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}" if rccl and os.environ.get("REMORA_DECOMP_SHARE_GPU", "") != "1" else "cuda:0"
    rem = REMORA(cfg, device=dev, box=((ilo, jlo, 0), (ihi, jhi, nz - 1)), nranks=(px, py), rank_xy=(rx, ry))
    if rccl:
        rem.comm_init(dist)
    else:
        rem.comm_set_host_transport(dist)
    ncalls = int(sys.argv[4]) if len(sys.argv) > 4 else -1          # debugging aid: stop after ncalls barotropic sub-steps of step 1

    def run(r):
        r.InitData()
        if land:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from landmask import put_masks_device
            put_masks_device(r, cfg, valid_only=land_valid_only)
        if ncalls < 0:
            r.Evolve(nsteps)
        else:
            r.begin_step(); r.setup_step()
            for q in range(ncalls):
                r.advance_2d_onestep(q)
        r.sync()
    if os.environ.get("REMORA_DECOMP_EXPECT_SPLIT", ""):
        rem.timing(True)
    run(rem)
    if os.environ.get("REMORA_DECOMP_EXPECT_SPLIT", ""):
        # the interior / rim split of k2d_fused is in force: two launches of the kernel per fused barotropic call
        rep = rem.timing_report(); rem.timing(False)
        per_step = 2 * rem.nfast                            # predictor and corrector of my_iif = 0 .. nfast - 1
        want = {"1": 2, "0": 1}[os.environ["REMORA_DECOMP_EXPECT_SPLIT"]] * per_step * nsteps
        assert rep["k2d_fused"]["calls"] == want, (rep["k2d_fused"], want)
        # include/remora_hip.h, rh_exchange_fn: the hook is handed the stream its exchange is ordered on.  With the split the
        # exchanges of the barotropic loop run on the library's side stream (not the context's compute stream, the null stream here)
        # while the interior tiles are queued on the compute stream; without it (and with the setup_step fills still on the side
        # stream) the loop's exchanges come on the compute stream.
        if not rccl:
            side = sum(1 for q in rem.hook_streams if q != 0)
            loop_calls = per_step * nsteps                  # at least one exchange phase per fused barotropic call
            if os.environ["REMORA_DECOMP_EXPECT_SPLIT"] == "1":
                assert side >= loop_calls, (side, loop_calls, len(rem.hook_streams))
            else:
                assert len(rem.hook_streams) - side >= loop_calls, (side, loop_calls, len(rem.hook_streams))
    mine = {}
    fields = FIELDS + (["tke", "gls", "Lscale", "Akk", "Akp", "Akt"] if "gls" in cfg else []) + (["mskp", "mskr", "msku", "mskv"] if land else [])
    for f in fields:
        a, lo, hi = rem.get(f)
        mine[f] = (a, lo, hi)
    gathered = [None] * world
    dist.gather_object((ilo, ihi, jlo, jhi, mine), gathered if rank == 0 else None, dst=0)
    # plotfile of the decomposed run: every rank appends its FAB records to its own data files, rank 0 writes the headers
    import ctypes as C
    import tempfile
    from remora_amd import lib as L
    pdir = [tempfile.mkdtemp(prefix="rh_plt_") if rank == 0 else None]
    dist.broadcast_object_list(pdir, src=0)
    pb = rem.write_plot_data(pdir[0], rank) if ncalls < 0 else None
    pbs = [None] * world
    dist.gather_object(bytes(pb) if pb is not None else None, pbs if rank == 0 else None, dst=0)
    ok = True
    if rank == 0:
        ref = REMORA(cfg, device=dev)
        run(ref)
        for f in fields:
            g, glo, ghi = ref.get(f)
            gng1 = -glo[1]
            nod_j = (ghi[1] - glo[1] + 1) - 2 * gng1 - ny
            gng0 = -glo[0]
            nod_i = (ghi[0] - glo[0] + 1) - 2 * gng0 - nx
            for (silo, sihi, sjlo, sjhi, fields) in gathered:
                a, lo, hi = fields[f]
                j0, j1 = sjlo, sjhi + nod_j                      # valid rows of this box's array
                i0, i1 = silo, sihi + nod_i                      # valid columns
                x = a[:, :, j0 - lo[1]:j1 - lo[1] + 1, i0 - lo[0]:i1 - lo[0] + 1]
                y = g[:, :, j0 - glo[1]:j1 - glo[1] + 1, i0 - glo[0]:i1 - glo[0] + 1]
                if f in ("rubar", "rvbar"):
                    # the stored barotropic right-hand side on a wall face is a don't-care: the ubar/vbar it feeds is overwritten
                    # there by the ext_dir boundary condition, and its value depends on ghost corners that FillBoundary never
                    # fills (in the reference's own multi-box runs as well).  Compare it away from non-periodic wall faces.
                    x = x.copy(); y = y.copy()
                    if f == "rubar" and not cfg["periodic"][0]:
                        if i0 == 0:
                            x[..., 0] = y[..., 0] = 0.0
                        if i1 == nx:
                            x[..., -1] = y[..., -1] = 0.0
                    if f == "rvbar" and not cfg["periodic"][1]:
                        if j0 == 0:
                            x[:, :, 0, :] = y[:, :, 0, :] = 0.0
                        if j1 == ny:
                            x[:, :, -1, :] = y[:, :, -1, :] = 0.0
                if not np.array_equal(x, y):
                    ok = False
                    w = np.argwhere(x != y)
                    print(f"MISMATCH {case} field {f} box i {silo}-{sihi} j {sjlo}-{sjhi}: max |diff| = {np.nanmax(np.abs(x - y)):.3e} at {len(w)} points; "
                          f"comps {sorted(set(int(v) for v in w[:, 0]))} i {sorted(set(int(v) for v in w[:, 3] + i0))[:12]} j {sorted(set(int(v) for v in w[:, 2] + j0))[:12]}", flush=True)
        if ncalls < 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from plotfile import read_plotfile
            B = (L.PlotBox * world)(*[L.PlotBox.from_buffer_copy(b) for b in pbs])
            L.check(rem.lib.rh_plotfile_write_headers(C.byref(rem.params), pdir[0].encode(), 0, rem.istep, rem.t_new, world, B))
            _, stitched = read_plotfile("", root=pdir[0])
            if not np.array_equal(stitched, ref.plot()):
                ok = False
                print(f"MISMATCH {case}: plotfile stitched from {world} boxes differs from the single-box plot variables", flush=True)
            import shutil
            shutil.rmtree(pdir[0], ignore_errors=True)
        print(("DECOMP_OK" if ok else "DECOMP_FAIL"), case, "world", world, "steps", nsteps, flush=True)
        ref.close()
    rem.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
