"""CPU tests that PIN the oracle: the C restatement under oracle/ run for the reference's own regression cases
(Tests/test_files/<case>/<case>.i, max_step = 10) must reproduce the reference's gold plotfiles
(Tests/REMORA_Gold_Files/<case>, committed as data under tests/golden/) under amrex_fcompare's criterion at the
reference's own tolerance, rel 1e-11 / abs 1e-11 (Tests/CTestList.cmake:39).

These are the only golden vectors the reference holds for the path (it has no per-function fixtures).
"""
import numpy as np
import pytest

from remora_amd import cases
from oracle.oracle_py import Oracle, set_weights
from plotfile import read_plotfile, fcompare


@pytest.mark.parametrize("case", ["Upwelling", "Seamount", "DoubleGyre", "Advection", "DoublyPeriodic", "DoublyPeriodic_bathy", "Upwelling_GLS", "Channel_Test"])
def test_oracle_reproduces_gold_plotfile(case):
    names, gold = read_plotfile(case)
    orc = Oracle(cases.get(case))
    orc.advance(10)
    mine = orc.plot(with_scalar=(len(names) == 6))
    orc.close()
    assert mine.shape == gold.shape
    res = fcompare(gold, mine)
    assert all(r[2] for r in res), f"{case}: {list(zip(names, res))}"


@pytest.mark.parametrize("case", ["Upwelling", "Seamount", "DoubleGyre", "Advection", "DoublyPeriodic", "DoublyPeriodic_bathy", "Upwelling_GLS", "Channel_Test"])
def test_gold_cell_h_minmax(case):
    """The per-box min/max of every component recorded in the gold Cell_H must match what the FAB reader returns: pins the
    reader (byte order, component order, multi-box assembly for Advection's 9 boxes)."""
    import os
    import re
    from plotfile import GOLD
    names, gold = read_plotfile(case)
    with open(os.path.join(GOLD, case, "Level_0", "Cell_H")) as f:
        txt = f.read()
    # two blocks "nbox,ncomp" followed by nbox rows of ncomp numbers (min block, then max block)
    blocks = re.findall(r"(\d+),(\d+)\n((?:[-+0-9.eE,]+\n)+)", txt)
    assert len(blocks) >= 2
    mins = np.array([[float(x) for x in row.strip(",").split(",")] for row in blocks[0][2].strip().split("\n")])
    maxs = np.array([[float(x) for x in row.strip(",").split(",")] for row in blocks[1][2].strip().split("\n")])
    for c in range(len(names)):
        assert np.isclose(gold[c].min(), mins[:, c].min(), rtol=1e-14, atol=0.0)
        assert np.isclose(gold[c].max(), maxs[:, c].max(), rtol=1e-14, atol=0.0)


@pytest.mark.parametrize("ndtfast", [20, 30])
def test_set_weights_properties(ndtfast):
    """REMORA::set_weights (TimeIntegration/REMORA_set_weights.cpp:10-171): the primary weights sum to 1 and have centroid
    ndtfast (the filter is centred on the new baroclinic time), the secondary weights sum to 1, nfast <= 2 ndtfast."""
    nfast, w1, w2 = set_weights(ndtfast)
    assert ndtfast <= nfast <= 2 * ndtfast
    assert abs(w1.sum() - 1.0) < 1e-13
    assert abs(w2.sum() - 1.0) < 1e-13
    # REMORA stores weight i at index i-1 ... the centroid of w1 over the 1-based fast-step index is ndtfast
    idx = np.arange(1, 2 * ndtfast + 2)
    assert abs((w1 * idx).sum() - ndtfast) < 1e-9 * ndtfast
    assert (w1[nfast:] == 0).all()


def test_threaded_oracle_is_bitwise_the_serial_one():
    """bench.py times the oracle with OpenMP threads over the leaf passes (oracle/fab.h: FORBOX); every parity test uses the
    serial order.  Both must give the same bits."""
    from remora_amd import cases
    from oracle.oracle_py import Oracle
    for name in ("Seamount", "DoubleGyre"):
        cfg = cases.get(name)
        a = Oracle(cfg); a.advance(3)
        b = Oracle(cfg, threads=4); b.advance(3)
        assert np.array_equal(a.plot(), b.plot()), name
        a.close(); b.close()
