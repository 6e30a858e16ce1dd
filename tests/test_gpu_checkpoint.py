"""Checkpoint / restart (REMORA::WriteCheckpointFile / ReadCheckpointFile, IO/REMORA_Checkpoint.cpp): a run interrupted after five
steps, written, read into a fresh context and continued for five more must equal the uninterrupted ten-step run BIT FOR BIT in
every state array the path carries from step to step, ghost cells included."""
import os

import numpy as np
import pytest

from remora_amd import cases, lib as L
from remora_amd.model import REMORA

pytestmark = pytest.mark.gpu

# work arrays that every step rebuilds before reading them are not part of the comparison (the checkpoint does not hold them)
CARRIED = ["cons_old", "cons_new", "xvel_old", "xvel_new", "yvel_old", "yvel_new", "zvel_old", "zvel_new", "Zt_avg1", "h", "ubar", "vbar",
           "ru", "rv", "ru2d", "rv2d", "z_w", "z_r", "Hz", "Huon", "Hvom", "Akv", "Akt", "DU_avg1", "DU_avg2", "DV_avg1", "DV_avg2", "zeta",
           "sstore", "rhoS", "rhoA", "bvf", "rufrc", "rvfrc", "bustr", "bvstr", "sustr", "svstr", "rubar", "rvbar", "rzeta"]


@pytest.mark.parametrize("case", ["Seamount", "DoubleGyre", "Upwelling_GLS"])
def test_restart_is_bit_identical(case, tmp_path):
    cfg = cases.get(case)
    a = REMORA(cfg); a.InitData(); a.Evolve(10)
    b = REMORA(cfg); b.InitData(); b.Evolve(5)
    chk = tmp_path / "chk00005"
    b.WriteCheckpointFile(chk)
    b.close()
    hdr = open(os.path.join(chk, "Header")).read().split("\n")
    assert hdr[0] == "Checkpoint file for REMORA" and hdr[10].split() == ["5"]
    assert os.path.exists(os.path.join(chk, "Level_0", "Cell_H")) and os.path.exists(os.path.join(chk, "Level_0", "XBar_D_00000"))
    c = REMORA(cfg)
    c.ReadCheckpointFile(chk)
    assert c.istep == 5 and c.t_new == 5 * cfg["fixed_dt"]
    c.Evolve(5)
    fields = CARRIED + (["tke", "gls", "Lscale", "Akk", "Akp"] if "gls" in cfg else [])
    bad = []
    for f in fields:
        x, _, _ = a.get(f); y, _, _ = c.get(f)
        if not np.array_equal(x, y):
            bad.append((f, float(np.nanmax(np.abs(x - y)))))
    assert not bad, f"{case}: restarted run differs: {bad}"
    assert np.array_equal(a.plot(), c.plot())
    a.close(); c.close()


def test_checkpoint_of_another_decomposition_is_refused(tmp_path):
    cfg = cases.get("Seamount")
    b = REMORA(cfg); b.InitData(); b.WriteCheckpointFile(tmp_path / "chk")
    b.close()
    small = dict(cfg, ncell=(40, 48, 13))
    c = REMORA(small)
    with pytest.raises(L.RemoraHipError, match="same decomposition"):
        c.ReadCheckpointFile(tmp_path / "chk")
    c.close()


# the VisMFs REMORA::WriteCheckpointFile writes for every run, GLS or not (IO/REMORA_Checkpoint.cpp:120-202), with the
# (components, ghost widths) the reference allocates them with (Initialization/REMORA_make_new_level.cpp:259-282,426-503)
# A native state carries no ghost cells in z (rh_state: "z ghost cells are optional"), so the 3-D state arrays are written with ghost
# widths (3,3,0) where the reference's MultiFabs have 3 in every direction.
REFERENCE_LIST = {"Cell": (3, "(3,3,0)"), "Cell_old": (3, "(3,3,0)"), "XFace": (1, "(3,3,0)"), "XFace_old": (1, "(3,3,0)"), "YFace": (1, "(3,3,0)"), "YFace_old": (1, "(3,3,0)"),
                  "ZFace": (1, "(3,3,0)"), "ZFace_old": (1, "(3,3,0)"), "XRHS": (2, "(2,2,0)"), "YRHS": (2, "(2,2,0)"),
                  "XBar": (3, "(2,2,0)"), "YBar": (3, "(2,2,0)"), "XRHS2d": (2, "(2,2,0)"), "YRHS2d": (2, "(2,2,0)"),
                  "Mskr": (1, "(3,3,0)"), "Msku": (1, "(3,3,0)"), "Mskv": (1, "(3,3,0)"), "rdrag": (1, "(2,2,0)"),
                  "DU_avg1": (1, "(2,2,0)"), "DU_avg2": (1, "(2,2,0)"), "DV_avg1": (1, "(2,2,0)"), "DV_avg2": (1, "(2,2,0)"),
                  "zeta": (3, "(3,3,0)"), "Zt_avg1": (1, "(3,3,0)"), "h": (2, "(3,3,0)"),
                  "tke": (3, "(2,2,0)"), "gls": (3, "(2,2,0)"), "Lscale": (1, "(2,2,0)"), "Akk": (1, "(2,2,0)"), "Akp": (1, "(2,2,0)"),
                  "Akv": (1, "(2,2,0)"), "Akt": (3, "(2,2,0)")}


@pytest.mark.parametrize("case", ["Seamount", "Upwelling_GLS"])
def test_checkpoint_holds_the_reference_file_list(case, tmp_path):
    """A checkpoint of a run WITHOUT the GLS closure must still hold tke / gls / Lscale / Akk / Akp: the reference reads them
    unconditionally (REMORA_Checkpoint.cpp:448-452).  Parity unpinned (the reference ships no checkpoint fixture): the list and
    shapes are checked against the reference's source."""
    cfg = cases.get(case)
    b = REMORA(cfg); b.InitData(); b.Evolve(1)
    chk = tmp_path / "chk00001"
    b.WriteCheckpointFile(chk)
    nx, ny, nz = cfg["ncell"]
    lev = os.path.join(chk, "Level_0")
    names = sorted(f[:-2] for f in os.listdir(lev) if f.endswith("_H"))
    assert names == sorted(REFERENCE_LIST), set(names) ^ set(REFERENCE_LIST)
    for name, (ncomp, ng) in REFERENCE_LIST.items():
        hdr = open(os.path.join(lev, name + "_H")).read().split("\n")
        assert hdr[0] == "1" and hdr[1] == "1" and int(hdr[2]) == ncomp, (name, hdr[:4])
        assert hdr[3] == ng, (name, hdr[3])
        assert os.path.getsize(os.path.join(lev, name + "_D_00000")) > 0
    if "gls" not in cfg:
        # what init leaves in a run without the closure: gls_Kmin, gls_Pmin, 0, Akk_bak, Akp_bak (make_new_level.cpp:517-521)
        for name, val, ncomp in (("tke", 7.6e-6, 3), ("gls", 1.0e-12, 3), ("Lscale", 0.0, 1), ("Akk", 5.0e-6, 1), ("Akp", 5.0e-6, 1)):
            raw = open(os.path.join(lev, name + "_D_00000"), "rb").read()
            head, body = raw.split(b"\n", 1)
            assert head.endswith(b"(0,0,1)) %d" % ncomp), head
            a = np.frombuffer(body, dtype="<f8")
            assert a.size == ncomp * (nx + 4) * (ny + 4) * (nz + 1) and np.all(a == val), name
    b.close()


def test_plotfile_rewrite_does_not_grow(tmp_path):
    """A second plotfile written into the same directory replaces the data files (one FAB record per rank and file)."""
    cfg = cases.get("Seamount")
    b = REMORA(cfg); b.InitData()
    d = tmp_path / "plt"
    b.WritePlotFile(d)
    n1 = os.path.getsize(os.path.join(d, "Level_0", "Cell_D_00000"))
    pb = b.write_plot_data(d, 0)
    assert pb.cell_offset == 0 and pb.nd_offset == 0
    assert os.path.getsize(os.path.join(d, "Level_0", "Cell_D_00000")) == n1
    b.close()


@pytest.mark.parametrize("case", ["Seamount", "DoubleGyre"])
def test_restart_with_land_is_bit_identical(case, tmp_path):
    """The same with land in the domain (tests/landmask.py).  The checkpoint holds mskr / msku / mskv as the reference's does
    (IO/REMORA_Checkpoint.cpp:172-182, 424-434) but not mskp, which the reference leaves at init_masks' ones on a restart; a caller that
    wants the run continued unchanged rebuilds it from the restored mskr (rh_update_mskp), as done here."""
    from landmask import put_masks_device
    cfg = dict(cases.get(case), masks_all_one=0)
    a = REMORA(cfg); a.InitData(); put_masks_device(a, cfg); a.Evolve(8)
    b = REMORA(cfg); b.InitData(); put_masks_device(b, cfg); b.Evolve(4)
    chk = tmp_path / "chk00004"
    b.WriteCheckpointFile(chk)
    b.close()
    c = REMORA(cfg)
    c.ReadCheckpointFile(chk)
    for m in ("mskr", "msku", "mskv"):
        assert np.array_equal(a.get(m)[0], c.get(m)[0]), m
    assert (c.get("mskp")[0] == 1.0).all()                       # what the reference's restart leaves
    c.update_mskp()
    assert np.array_equal(a.get("mskp")[0], c.get("mskp")[0])
    c.Evolve(4)
    bad = [f for f in CARRIED if not np.array_equal(a.get(f)[0], c.get(f)[0])]
    assert not bad, f"{case}: restarted run with land differs in {bad}"
    a.close(); c.close()
