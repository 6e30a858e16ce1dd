"""CPU test of the multi-rank path (world_size 2 and 3, gloo): the exchange geometry rh_halo_plan_j gives each rank
(host-only entry point of the C ABI), driven with numpy arrays over torch.distributed, must fill every ghost row exactly as
a FillBoundary of the undecomposed array would (periodic wrap included; two ranks + periodic = both neighbours are one peer)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _slab(ny, world, rank):
    base, rem = divmod(ny, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0) - 1


def _gval(i, j, k, nx, ny, perx, pery, nod_i, nod_j):
    """value of the undecomposed field at global index (i,j,k); NaN where FillBoundary has no source.  Points inside the
    global valid node range keep their own value (for a nodal array the node rows 0 and n are both valid); only points
    outside it wrap."""
    ii, jj = i, j
    if perx and not (0 <= i <= nx - 1 + nod_i):
        ii = i % nx
    if pery and not (0 <= j <= ny - 1 + nod_j):
        jj = j % ny
    if ii < 0 or ii > nx - 1 + nod_i or jj < 0 or jj > ny - 1 + nod_j:
        return np.nan
    return 1000.0 * k + ii + 0.001 * jj


def _worker(rank, world, port, nx, ny, nz, perx, pery, ng, nod_i, nod_j, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from remora_amd import lib as L
    lib = L.load()
    jlo, jhi = _slab(ny, world, rank)
    p = L.Params()
    for d, n in enumerate((nx, ny, nz)):
        p.domain_lo[d] = 0; p.domain_hi[d] = n - 1
    p.box_lo[0] = 0; p.box_hi[0] = nx - 1; p.box_lo[1] = jlo; p.box_hi[1] = jhi; p.box_lo[2] = 0; p.box_hi[2] = nz - 1
    p.periodic[0] = perx; p.periodic[1] = pery; p.nranks[0] = 1; p.nranks[1] = world; p.rank_xy[0] = 0; p.rank_xy[1] = rank
    alo = (C.c_int * 3)(-ng, jlo - ng, 0); ahi = (C.c_int * 3)(nx - 1 + nod_i + ng, jhi + nod_j + ng, nz - 1)
    ngv = (C.c_int * 3)(ng, ng, 0)
    pl = L.HaloPlan()
    rc = lib.rh_halo_plan_j(C.byref(p), alo, ahi, ngv, C.byref(pl))
    assert rc == 0, lib.rh_last_error()
    NI, NJ = ahi[0] - alo[0] + 1, ahi[1] - alo[1] + 1
    a = np.full((nz, NJ, NI), np.nan)
    for k in range(nz):                                   # valid region
        for j in range(jlo, jhi + nod_j + 1):
            for i in range(0, nx + nod_i):
                a[k, j - alo[1], i - alo[0]] = _gval(i, j, k, nx, ny, 0, 0, nod_i, nod_j)
    if perx:                                              # local x wrap of the valid rows (what the device kernel does first)
        for j in range(jlo, jhi + nod_j + 1):
            for i in list(range(alo[0], 0)) + list(range(nx + nod_i, ahi[0] + 1)):
                a[:, j - alo[1], i - alo[0]] = a[:, j - alo[1], (i % nx) - alo[0]]
    i0, i1 = pl.i_lo - alo[0], pl.i_hi - alo[0] + 1

    def rows(j0):
        return torch.from_numpy(np.ascontiguousarray(a[:, j0 - alo[1]:j0 - alo[1] + pl.nrows, i0:i1]))
    reqs, recvs = [], []
    for peer, j0, tag in ((pl.peer_lo, pl.send_lo_j, 0), (pl.peer_hi, pl.send_hi_j, 1)):
        if peer >= 0:
            reqs.append(dist.isend(rows(j0), dst=peer, tag=tag))
    for peer, j0, tag in ((pl.peer_hi, pl.recv_hi_j, 0), (pl.peer_lo, pl.recv_lo_j, 1)):
        if peer >= 0:
            t = torch.empty((nz, pl.nrows, i1 - i0), dtype=torch.float64)
            reqs.append(dist.irecv(t, src=peer, tag=tag)); recvs.append((j0, t))
    for r in reqs:
        r.wait()
    for j0, t in recvs:
        a[:, j0 - alo[1]:j0 - alo[1] + pl.nrows, i0:i1] = t.numpy()
    # expectation: the undecomposed FillBoundary
    bad = 0
    for k in (0, nz - 1):
        for j in range(alo[1], ahi[1] + 1):
            for i in range(alo[0], ahi[0] + 1):
                e = _gval(i, j, k, nx, ny, perx, pery, nod_i, nod_j)
                g = a[k, j - alo[1], i - alo[0]]
                if np.isnan(e):
                    bad += 0 if np.isnan(g) else 1      # nothing may be written where FillBoundary has no source
                else:
                    bad += 0 if g == e else 1
    q.put((rank, bad))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,perx,pery,nod_i,nod_j", [(2, 1, 1, 0, 0), (2, 1, 1, 0, 1), (2, 1, 0, 1, 0), (3, 0, 0, 0, 1), (3, 0, 1, 0, 0), (2, 0, 0, 0, 0)])
def test_plan_exchange_equals_fill_boundary(world, perx, pery, nod_i, nod_j):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + world * 16 + perx * 8 + pery * 4 + nod_i * 2 + nod_j
    procs = [ctx.Process(target=_worker, args=(r, world, port, 12, 17, 3, perx, pery, 3, nod_i, nod_j, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(b == 0 for _, b in res), res


# ---- two-phase exchange on an i-j process grid (phase X: rh_halo_plan_i, phase Y: rh_halo_plan_j) ---------------------------
def _worker_ij(rank, px, py, port, nx, ny, nz, perx, pery, ng, nod_i, nod_j, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    world = px * py
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from remora_amd import lib as L
    lib = L.load()
    rx, ry = rank % px, rank // px
    ilo, ihi = _slab(nx, px, rx)
    jlo, jhi = _slab(ny, py, ry)
    p = L.Params()
    for d, n in enumerate((nx, ny, nz)):
        p.domain_lo[d] = 0; p.domain_hi[d] = n - 1
    p.box_lo[0] = ilo; p.box_hi[0] = ihi; p.box_lo[1] = jlo; p.box_hi[1] = jhi; p.box_lo[2] = 0; p.box_hi[2] = nz - 1
    p.periodic[0] = perx; p.periodic[1] = pery; p.nranks[0] = px; p.nranks[1] = py; p.rank_xy[0] = rx; p.rank_xy[1] = ry
    alo = (C.c_int * 3)(ilo - ng, jlo - ng, 0); ahi = (C.c_int * 3)(ihi + nod_i + ng, jhi + nod_j + ng, nz - 1)
    ngv = (C.c_int * 3)(ng, ng, 0)
    NI, NJ = ahi[0] - alo[0] + 1, ahi[1] - alo[1] + 1
    a = np.full((nz, NJ, NI), np.nan)
    for k in range(nz):
        for j in range(jlo, jhi + nod_j + 1):
            for i in range(ilo, ihi + nod_i + 1):
                a[k, j - alo[1], i - alo[0]] = _gval(i, j, k, nx, ny, 0, 0, nod_i, nod_j)

    def local_wrap(d, n, nod, lo_idx):
        # a direction owned by one rank wraps locally (only reached with px == 1 or py == 1)
        valid_hi = n - 1 + nod
        for g in list(range(lo_idx, 0)) + list(range(valid_hi + 1, valid_hi + ng + 1)):
            src = g % n
            if d == 0:
                a[:, jlo - alo[1]:jhi + nod_j - alo[1] + 1, g - alo[0]] = a[:, jlo - alo[1]:jhi + nod_j - alo[1] + 1, src - alo[0]]
            else:
                a[:, g - alo[1], :] = a[:, src - alo[1], :]

    def phase(d):
        pl = L.HaloPlan()
        fn = lib.rh_halo_plan_i if d == 0 else lib.rh_halo_plan_j
        assert fn(C.byref(p), alo, ahi, ngv, C.byref(pl)) == 0, lib.rh_last_error()
        o0, o1 = pl.i_lo - alo[1 - d], pl.i_hi - alo[1 - d] + 1

        def blk(first):
            f = first - alo[d]
            return a[:, o0:o1, f:f + pl.nrows] if d == 0 else a[:, f:f + pl.nrows, o0:o1]
        reqs, recvs = [], []
        for peer, first, tag in ((pl.peer_lo, pl.send_lo_j, 0), (pl.peer_hi, pl.send_hi_j, 1)):
            if peer >= 0:
                reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(blk(first))), dst=peer, tag=tag))
        for peer, first, tag in ((pl.peer_hi, pl.recv_hi_j, 0), (pl.peer_lo, pl.recv_lo_j, 1)):
            if peer >= 0:
                t = torch.empty(blk(first).shape, dtype=torch.float64)
                reqs.append(dist.irecv(t, src=peer, tag=tag)); recvs.append((first, t))
        for r in reqs:
            r.wait()
        for first, t in recvs:
            blk(first)[...] = t.numpy()

    if px == 1 and perx:
        local_wrap(0, nx, nod_i, alo[0])
    if px > 1:
        phase(0)
    if py > 1:
        phase(1)
    elif pery:
        # rows wrap over the columns FillBoundary fills: restrict to them by restoring NaN elsewhere afterwards
        keep = a.copy()
        local_wrap(1, ny, nod_j, alo[1])
        left = perx or rx > 0; right = perx or rx < px - 1
        if not left:
            a[:, :, :ng] = keep[:, :, :ng]
        if not right:
            a[:, :, NI - ng:] = keep[:, :, NI - ng:]
    bad = 0
    for k in (0, nz - 1):
        for j in range(alo[1], ahi[1] + 1):
            for i in range(alo[0], ahi[0] + 1):
                e = _gval(i, j, k, nx, ny, perx, pery, nod_i, nod_j)
                # this rank's own valid points keep their own value even where the periodic image differs (shared nodes)
                g = a[k, j - alo[1], i - alo[0]]
                if np.isnan(e):
                    bad += 0 if np.isnan(g) else 1
                else:
                    bad += 0 if g == e else 1
    q.put((rank, bad))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("px,py,perx,pery,nod_i,nod_j", [(2, 2, 1, 1, 0, 0), (2, 2, 0, 0, 1, 0), (2, 2, 1, 0, 0, 1), (2, 1, 1, 1, 1, 0), (3, 1, 0, 1, 0, 0),
                                                         (2, 2, 0, 1, 0, 0)])
def test_two_phase_exchange_equals_fill_boundary(px, py, perx, pery, nod_i, nod_j):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + px * 32 + py * 16 + perx * 8 + pery * 4 + nod_i * 2 + nod_j
    procs = [ctx.Process(target=_worker_ij, args=(r, px, py, port, 13, 17, 2, perx, pery, 3, nod_i, nod_j, q)) for r in range(px * py)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(b == 0 for _, b in res), res
