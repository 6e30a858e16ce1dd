"""Reader for the AMReX native plotfile FABs kept under tests/golden/ (format: SURVEY.md section 4),
and the norm-wise comparison amrex_fcompare applies (Tests/CTestList.cmake:39: -r 1e-11 --abs_tol 1e-11).  The code lives in the
package (remora_amd/plotfile.py; tools/rh_fcompare.py is its command line); this module only adds the gold directory."""
import os

from remora_amd.plotfile import read_plotfile_dir, fcompare      # noqa: F401

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_plotfile(case, root=None, nodal=False):
    """Cell data [comp, k, j, i] of a plotfile directory (default: tests/golden/<case>); nodal=True reads the Nu_nd block."""
    return read_plotfile_dir(os.path.join(GOLD if root is None else root, case), nodal)
