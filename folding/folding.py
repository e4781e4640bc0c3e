#!/usr/bin/env python
"""Command-line shim with the flags of the reference's fold script (/root/reference/folding/folding.py +
folding/utils_ros/arguments.py:5-25):  python ./folding/folding.py -NPZ x.npz -FASTA s.fasta -OUT o.pdb [-m 2]
[-r no-idp] [--orient|--no-orient] [--fastrelax|--no-fastrelax] [-pd P] ...   One decoy, folded on the GPU; exit code 0
on success.  --fastrelax (the default, as in the reference) appends the backbone-visible part of the full-atom refinement
(protocol.relax_runs; no side chains: DESIGN.md section 2), --no-fastrelax skips it; there is no CPU fallback."""
import argparse
import importlib
import os
import sys
from time import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    ap.add_argument("-NPZ", type=str, required=True, help="input distograms and anglegrams (NN predictions)")
    ap.add_argument("-FASTA", type=str, required=True, help="input sequence")
    ap.add_argument("-OUT", type=str, required=True, help="output model (in PDB format)")
    ap.add_argument("--seed", type=int, default=None, help="seed of the random start torsions (extension)")
    ap.add_argument("--device", type=int, default=0, help="GPU index (extension)")
    args, rest = ap.parse_known_args(argv)
    fold = importlib.import_module("trrosettax2-dynamics_amd.fold")
    s = time()
    fold.fold_npz(args.NPZ, args.FASTA, args.OUT, rest, device=args.device, seed=args.seed)
    print("\ndone")
    print(f"*** time:{time() - s:.2f}s ***")
    return 0


if __name__ == "__main__":
    sys.exit(main())
