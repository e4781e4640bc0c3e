"""Golden DATA: the per-residue energy tables the reference's eight committed decoys carry (#BEGIN_POSE_ENERGIES_TABLE, written by
pose.dump_pdb, folding/folding.py:273) -- the only energy-level numbers of Rosetta's the reference tree holds.
Run in the build container (reads /root/reference/example/output/seq/pred_pdb/conf_*.pdb); writes tests/golden/pose_energies.json:
per decoy the provenance line, the column labels, ref2015_cart's WEIGHTS as Rosetta printed them, the pose totals and, per residue,
the weighted columns a backbone model has a counterpart for (omega, rama_prepro, hbond_sr_bb, hbond_lr_bb, cart_bonded, fa_rep).
Numbers only; no reference source text."""
import glob
import json
import os

KEEP = ("fa_rep", "hbond_sr_bb", "hbond_lr_bb", "omega", "rama_prepro", "cart_bonded", "total")
SRC = "/root/reference/example/output/seq/pred_pdb"
out = {}
for path in sorted(glob.glob(os.path.join(SRC, "conf_*.pdb"))):
    lines = open(path).read().splitlines()
    i0 = next(i for i, l in enumerate(lines) if l.startswith("#BEGIN_POSE_ENERGIES_TABLE"))
    i1 = next(i for i, l in enumerate(lines) if l.startswith("#END_POSE_ENERGIES_TABLE"))
    labels = lines[i0 + 1].split()[1:]
    weights = dict(zip(labels, lines[i0 + 2].split()[1:]))
    pose = dict(zip(labels, map(float, lines[i0 + 3].split()[1:])))
    res = [l.split() for l in lines[i0 + 4:i1]]
    assert len(res) == 90 and all(len(r) == len(labels) + 1 for r in res)
    col = {k: labels.index(k) + 1 for k in KEEP}
    out[os.path.basename(path)[:-4]] = {
        "provenance": lines[i0].split(None, 1)[1], "weights": {k: (None if v == "NA" else float(v)) for k, v in weights.items()},
        "pose": pose, "residues": [r[0] for r in res], "per_residue": {k: [float(r[c]) for r in res] for k, c in col.items()}}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pose_energies.json"), "w"), indent=0)
print({k: (v["provenance"], v["pose"]["total"]) for k, v in out.items()})
