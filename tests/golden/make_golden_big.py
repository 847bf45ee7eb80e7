#!/usr/bin/env python3
"""Golden vectors above the one-wavefront kernels' 64-variable limit, from the reference itself (build container
only; same accommodation as make_golden.py): its own profiler shape nz = nineq = 100, neq = 0
(prof-linear.py:38-46,64-75) and a case with equality rows (nz 100, nineq 60, neq 30), QPFunction forward +
backward (two cotangents).  (DenseQPFunction is not recorded at these sizes: its preprocess builds the full KKT
matrix with Python loops over the batch and the 250 rows.)

Usage:  python tests/golden/make_golden_big.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402  (installs the ipdb stand-in, imports the reference)


def main():
    d = mg.family_R(seed=11, B=3, nz=100, nineq=100, neq=0)
    mg.save("R_prof100_b3", d, mg.run_qpfunction(d))
    d = mg.family_R(seed=12, B=3, nz=100, nineq=60, neq=30)
    mg.save("R_big_eq_b3", d, mg.run_qpfunction(d))


if __name__ == "__main__":
    main()
