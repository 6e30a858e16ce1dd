"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol include/remora_hip.h
declares (and the ctypes table binds exactly that set), the host-only entry points work, and the product path refuses to
run without a device instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    with open(os.path.join(ROOT, "include", "remora_hip.h")) as f:
        txt = f.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(rh_[a-z0-9_]+)\s*\(", txt))
    names.discard("rh_exchange_fn")
    return names


def test_header_symbols_are_exported_and_bound():
    from remora_amd import lib as L
    lib = L.load()
    decl = _declared()
    assert len(decl) >= 40
    for n in sorted(decl):
        assert hasattr(lib, n), f"{n} declared in include/remora_hip.h but not exported"
    assert decl == set(L.SIGNATURES), (decl ^ set(L.SIGNATURES))


def test_struct_layouts_match_header():
    """ctypes mirrors of the POD structs: field order and count follow the header text."""
    from remora_amd import lib as L
    with open(os.path.join(ROOT, "include", "remora_hip.h")) as f:
        txt = f.read()
    body = re.search(r"typedef struct rh_state \{(.*?)\} rh_state;", txt, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)                     # comments inside the struct are not fields
    fields = re.findall(r"\b([A-Za-z_0-9]+)\s*[,;]", re.sub(r"rh_array4", "", body))
    assert fields == L.STATE_FIELDS
    assert C.sizeof(L.Array4) == 40 and C.sizeof(L.Box) == 24
    assert C.sizeof(L.State) == 40 * len(L.STATE_FIELDS)


def test_halo_plan_host_entry_point():
    """rh_halo_plan_j needs no device: 3 ranks, periodic in y, cell-centred array with 3 ghost rows."""
    from remora_amd import lib as L
    lib = L.load()
    p = L.Params()
    nx, ny, nz = 10, 30, 4
    for d, n in enumerate((nx, ny, nz)):
        p.domain_lo[d] = 0; p.domain_hi[d] = n - 1
    p.periodic[0] = 0; p.periodic[1] = 1; p.nranks[0] = 1; p.nranks[1] = 3
    plans = []
    for r in range(3):
        p.rank_xy[1] = r
        p.box_lo[0] = 0; p.box_hi[0] = nx - 1; p.box_lo[1] = 10 * r; p.box_hi[1] = 10 * r + 9; p.box_lo[2] = 0; p.box_hi[2] = nz - 1
        pl = L.HaloPlan()
        alo = (C.c_int * 3)(-3, 10 * r - 3, 0); ahi = (C.c_int * 3)(nx + 2, 10 * r + 12, nz - 1); ng = (C.c_int * 3)(3, 3, 0)
        assert lib.rh_halo_plan_j(C.byref(p), alo, ahi, ng, C.byref(pl)) == 0, lib.rh_last_error()
        plans.append(pl)
    assert [pl.peer_lo for pl in plans] == [2, 0, 1]
    assert [pl.peer_hi for pl in plans] == [1, 2, 0]
    for r, pl in enumerate(plans):
        assert pl.nrows == 3
        assert (pl.i_lo, pl.i_hi) == (0, nx - 1)                  # x is not periodic: no x ghost columns travel
        assert pl.send_lo_j == 10 * r and pl.send_hi_j == 10 * r + 7
        assert pl.recv_lo_j == 10 * r - 3 and pl.recv_hi_j == 10 * r + 10


def test_no_cpu_fallback():
    """Without a HIP device the host mirror must raise, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from remora_amd import cases, lib as L
    from remora_amd.model import REMORA
    with pytest.raises(L.RemoraHipError):
        REMORA(cases.get("Seamount"))


def test_product_does_not_import_oracle():
    """Nothing under remora_amd/ may import, link or name the oracle (it is test infrastructure)."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "remora_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", "Makefile")):
                with open(os.path.join(dp, f), errors="replace") as fh:
                    t = fh.read()
                assert "liboracle" not in t and "oracle_py" not in t and "import oracle" not in t and "from oracle" not in t, f
