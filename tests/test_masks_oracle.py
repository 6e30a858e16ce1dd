"""CPU checks of the oracle's land / sea mask helper (no GPU): o_update_mskp against an independent statement of the rule of
REMORA::update_mskp (Initialization/REMORA_make_new_level.cpp:617-655): count the wet cells around a node."""
import numpy as np

from remora_amd import cases
from oracle.oracle_py import Oracle


def test_update_mskp_rule_and_periodic_ghosts():
    cfg = cases.get("Seamount")                      # doubly periodic: the ghost cells of all four masks wrap around
    orc = Oracle(cfg)
    nx, ny = cfg["ncell"][0], cfg["ncell"][1]
    rng = np.random.default_rng(7)
    land = rng.random((ny, nx)) < 0.3
    r, lo, hi = orc.field("mskr")
    j, i = np.mgrid[lo[1]:hi[1] + 1, lo[0]:hi[0] + 1]
    r[0, 0] = np.where(land[np.mod(j, ny), np.mod(i, nx)], 0.0, 1.0)
    orc.update_mskp()
    p, plo, phi = orc.field("mskp")
    seen = set()
    for jj in range(-1, ny + 2):
        for ii in range(-1, nx + 2):
            wet = [not land[(jj + dj) % ny, (ii + di) % nx] for dj, di in ((0, -1), (0, 0), (-1, -1), (-1, 0))]
            n = sum(wet)
            if n >= 3:
                want = 1.0
            elif n == 2 and (wet[0] == wet[2] or wet[0] == wet[1]):     # the two dry cells share an edge: a straight coast
                want = 2.0
            else:
                want = 0.0
            seen.add(want)
            assert p[0, 0, jj - plo[1], ii - plo[0]] == want, (ii, jj, wet)
    assert seen == {0.0, 1.0, 2.0}
    # ghost nodes are periodic images of valid nodes (nodal arrays: node nx coincides with node 0)
    for jj in range(plo[1], phi[1] + 1):
        for ii in range(plo[0], phi[0] + 1):
            assert p[0, 0, jj - plo[1], ii - plo[0]] == p[0, 0, (jj % ny) - plo[1], (ii % nx) - plo[0]], (ii, jj)
    orc.close()


def test_timing_build_of_the_oracle_is_current():
    """bench.py's cpu_baseline leg loads oracle/liboracle_fast.so as prebuilt on the GPU box: it must export everything the Python
    binding declares, i.e. be built from the same sources as the parity build (__graft_entry__.build makes both)."""
    import ctypes as C
    import os
    from oracle import oracle_py
    fast = C.CDLL(os.path.join(os.path.dirname(oracle_py.__file__), "liboracle_fast.so"))
    for name in ("o_create", "o_destroy", "o_advance", "o_setup_step", "o_advance_2d_onestep", "o_advance_3d_ml", "o_begin_step", "o_end_step",
                 "o_update_mskp", "o_nfast", "o_istep", "o_field", "o_plot", "o_est_timestep", "o_sum_iq", "o_set_weights", "o_physbc_calc",
                 "o_fillpatch_calc", "o_get_bc", "o_init_bcs"):
        assert hasattr(fast, name), name
