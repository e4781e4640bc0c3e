"""CPU: the parts of the build's own PyRosetta driver (tools/pyrosetta_driver.py, the CPU leg of BASELINE.md section 3) that run
without PyRosetta: the SPLINE files and constraint lines it hands to Rosetta, and its add_rst selection -- pinned to the reference's
counts on the example map.  The Rosetta part cannot run here (no host seen so far has PyRosetta); bench.py runs it where it can."""
import importlib
import os
import sys

import numpy as np

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
D = importlib.import_module("tools.pyrosetta_driver")


def test_spline_files_constraint_lines_and_selection(golden_dir, seq, tmp_path):
    m = np.load(os.path.join(golden_dir, "seq_NMR.npz"))
    T = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    kn = T.knots()
    tab = {"gen": T.mask(False)}
    for ch in D.CH:
        tab[f"{ch}_x"], tab[f"{ch}_y"], tab[f"{ch}_p"] = kn[ch], T.y(ch), T.prob(ch)
    rst = D.write_restraints(tab, str(tmp_path))
    # gen_rst's generated counts on this map: 3226 / 3201 / 6455 / 4773
    assert [len(rst[ch]) for ch in D.CH] == [3226, 3201, 6455, 4773]
    # add_rst at PCUT 0.05 (utils_ros.py:719-723): 3226 / 2562 / 5142 / 2541 selected
    n_sel = [sum(1 for a, b, p, _ in rst[ch] if 1 <= abs(a - b) < 90 and p >= 0.05 + D.P_EXTRA[ch]) for ch in D.CH]
    assert n_sel == [3226, 2562, 5142, 2541]
    a, b, p, line = rst["dist"][0]
    w = line.split()
    assert w[:3] == ["AtomPair", "CB", str(a + 1)] and w[3:7] == ["CB", str(b + 1), "SPLINE", "TAG"] and w[-3:] == ["1.0", "1.000", "0.50000"]
    x_line, y_line = open(w[7]).read().splitlines()
    assert x_line.split("\t")[:4] == ["x_axis", "0.000", "2.000", "3.500"] and len(y_line.split("\t")) == 36
    assert rst["theta"][0][3].startswith("Dihedral N ") and rst["phi"][0][3].startswith("Angle CA ") and rst["omega"][0][3].startswith("Dihedral CA ")
    assert all(a < b for a, b, _, _ in rst["dist"]) and any(a > b for a, b, _, _ in rst["theta"])
