#!/usr/bin/env python3
"""rh_fcompare.py -- compare two AMReX-native plotfile directories the way the reference's regression tests do
(Tests/CTestList.cmake:39: `amrex_fcompare -a -r 1e-11 --abs_tol 1e-11 <gold> <plt00010>`).

    python tools/rh_fcompare.py [-a] [-r REL_TOL] [--abs_tol ABS_TOL] [--nodal] dirA dirB

Prints one line per variable (max-norm absolute error, error relative to dirA's max-norm) and exits 0 when every variable is within
either tolerance (both default to 0: bitwise-equal data, like amrex_fcompare), 1 otherwise, 2 on unreadable input.  -a /
--allow_diff_grids is accepted for command-line compatibility: box layouts never matter here (boxes are stitched into the level-0
domain before comparing).  --nodal compares the nodal mesh-height block (Nu_nd) instead of the cell data.  Host-only."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from remora_amd.plotfile import fcompare_dirs          # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(description="norm-wise comparison of two AMReX plotfiles (amrex_fcompare's criterion)")
    ap.add_argument("-a", "--allow_diff_grids", action="store_true", help="accepted for compatibility; layouts never matter here")
    ap.add_argument("-r", "--rel_tol", type=float, default=0.0)
    ap.add_argument("--abs_tol", type=float, default=0.0)
    ap.add_argument("--nodal", action="store_true", help="compare the nodal Nu_nd block instead of the cell data")
    ap.add_argument("dirA"); ap.add_argument("dirB")
    a = ap.parse_args(argv)
    try:
        ok, rows, problems = fcompare_dirs(a.dirA, a.dirB, a.rel_tol, a.abs_tol, a.nodal)
    except (OSError, ValueError, IndexError) as e:
        print(f"rh_fcompare: {e}", file=sys.stderr)
        return 2
    print(f"{'variable name':<24}{'absolute error':>24}{'relative error':>24}")
    print(f"{'':<24}{'(||A - B||)':>24}{'(||A - B||/||A||)':>24}")
    print("-" * 72)
    print(" level = 0")
    for name, aerr, rerr, good in rows:
        print(f" {name:<23}{aerr:>24.10e}{rerr:>24.10e}{'' if good else '   <-- beyond tolerance'}")
    for p in problems:
        print(" " + p)
    if ok and all(r[1] == 0.0 for r in rows):
        print("PLOTFILES AGREE (bitwise)")
    elif ok:
        print("PLOTFILES AGREE to specified tolerances")
    else:
        print("PLOTFILES DISAGREE")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
