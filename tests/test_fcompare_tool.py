"""tools/rh_fcompare.py, the fcompare-equivalent the reference's regression tests rely on (Tests/CTestList.cmake:39), on the gold
plotfiles themselves: a plotfile agrees with itself bitwise; two different cases of one domain disagree; a copy with one value
perturbed passes or fails exactly at the tolerance; the nine-box Advection gold stitches into its domain (-a).  The GPU side
(tests/test_gpu_plotfile.py) runs the same tool on a plotfile written by rh_write_plotfile against the gold directory."""
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TOOL = os.path.join(ROOT, "tools", "rh_fcompare.py")


def _run(*args):
    r = subprocess.run([sys.executable, TOOL] + list(args), capture_output=True, text=True, timeout=120)
    return r.returncode, r.stdout + r.stderr


def test_gold_agrees_with_itself_and_differs_from_another_case():
    rc, out = _run(os.path.join(GOLD, "Upwelling"), os.path.join(GOLD, "Upwelling"))
    assert rc == 0 and "AGREE (bitwise)" in out, out
    for v in ("temp", "salt", "x_velocity", "y_velocity", "z_velocity"):
        assert v in out
    # Upwelling_GLS: same domain and variables, other physics
    rc, out = _run("-a", "-r", "1e-11", "--abs_tol", "1e-11", os.path.join(GOLD, "Upwelling"), os.path.join(GOLD, "Upwelling_GLS"))
    assert rc == 1 and "DISAGREE" in out, out
    # different domains
    rc, out = _run(os.path.join(GOLD, "Upwelling"), os.path.join(GOLD, "Seamount"))
    assert rc == 1 and "domains differ" in out, out
    rc, out = _run(os.path.join(GOLD, "Upwelling"), os.path.join(GOLD, "no_such_dir"))
    assert rc == 2, out


def test_tolerance_is_the_reference_criterion(tmp_path):
    src = os.path.join(GOLD, "Seamount")
    dst = str(tmp_path / "plt")
    shutil.copytree(src, dst)
    fn = os.path.join(dst, "Level_0", "Cell_D_00000")
    raw = bytearray(open(fn, "rb").read())
    hdr = raw.index(b"\n") + 1
    a = np.frombuffer(raw, dtype="<f8", offset=hdr).copy()
    i = int(np.argmax(np.abs(a[:1000])))                 # a temperature value of the first rows
    a[i] *= 1.0 + 4e-12                                   # relative change 4e-12 of a value near the component's maximum
    raw[hdr:] = a.tobytes()
    open(fn, "wb").write(bytes(raw))
    rc, out = _run("-a", "-r", "1e-11", "--abs_tol", "1e-11", src, dst)
    assert rc == 0 and "AGREE to specified tolerances" in out, out
    rc, out = _run("-r", "1e-13", "--abs_tol", "1e-13", src, dst)
    assert rc == 1 and "beyond tolerance" in out, out
    rc, out = _run(src, dst)                              # default tolerances are zero, as amrex_fcompare's
    assert rc == 1, out


def test_nine_box_gold():
    """the Advection gold is nine boxes in four data files; the nodal block (--nodal) has no data file among the gold fixtures
    (only Nu_nd_H travels): asking for it is an input error, not a comparison"""
    rc, out = _run("-a", os.path.join(GOLD, "Advection"), os.path.join(GOLD, "Advection"))
    assert rc == 0 and "scalar" in out, out
    rc, out = _run("--nodal", os.path.join(GOLD, "Upwelling"), os.path.join(GOLD, "Upwelling"))
    assert rc == 2, out
