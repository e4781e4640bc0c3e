#!/usr/bin/env python
"""Same command line as the reference's run_inference.py (:356-390), folding on the GPU.

The trX2 network front-end is not part of this package, so --msa / --msa_dir are accepted for compatibility but the
distograms are read from {save_dir}/{name}/pred_npz/{name}_NMR.npz (_Xray.npz) or from --npz_nmr / --npz_xray."""
import argparse
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(argv=None):
    p = argparse.ArgumentParser(description="Protein conformation sampling: trRosettaX2-Dynamics fold on MI355X.")
    p.add_argument("--fasta", type=str); p.add_argument("--msa", type=str)
    p.add_argument("--fasta_dir", type=str); p.add_argument("--msa_dir", type=str)
    p.add_argument("--name", type=str); p.add_argument("--name_lst", type=str)
    p.add_argument("--save_dir", type=str, required=True)
    p.add_argument("--init_num", type=int, default=10)
    p.add_argument("--Nmax", type=int, default=300)
    p.add_argument("--angle", action=argparse.BooleanOptionalAction, default=True)
    p.add_argument("--mult_two_models", action=argparse.BooleanOptionalAction, default=True)
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--npz_nmr", type=str, default=None, help="precomputed NMR-model distogram (extension)")
    p.add_argument("--npz_xray", type=str, default=None, help="precomputed X-ray-model distogram (extension)")
    p.add_argument("--seed", type=int, default=None, help="seed of the random start torsions (extension)")
    a = p.parse_args(argv)
    if a.name_lst:
        if not a.fasta_dir:
            p.error("Batch mode requires --fasta_dir and --name_lst.")
    elif not a.fasta or not a.name:
        p.error("Single mode requires --fasta and --name.")
    dev = int(a.device.split(":")[1]) if ":" in a.device else 0
    pipe = importlib.import_module("trrosettax2-dynamics_amd.pipeline")
    kw = dict(init_num=a.init_num, Nmax=a.Nmax, angle=a.angle, mult_two_models=a.mult_two_models, device=dev, seed=a.seed)
    if a.name_lst:
        for name in [l.strip() for l in open(a.name_lst) if l.strip()]:     # run_inference.py:343-348
            pipe.run_single(name, os.path.join(a.fasta_dir, name + ".fasta"), a.save_dir, **kw)
    else:
        pipe.run_single(a.name, a.fasta, a.save_dir, npz_nmr=a.npz_nmr, npz_xray=a.npz_xray, **kw)
    return 0


if __name__ == "__main__":
    sys.exit(main())
