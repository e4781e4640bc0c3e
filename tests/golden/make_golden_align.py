"""Golden vectors for `evaluate.py --align` (TM-score's -seq option, /root/reference/utils_trX2dy/evaluate_utils.py:56-58).

Run in the BUILD container only: executes the reference's prebuilt /root/reference/bin/TMscore (a copy made executable under
/tmp) on pairs of PDB files derived from the reference's example natives (example/apo.pdb, example/holo.pdb) by random
deletions, point mutations and renumbering, and records for every pair the two sequences, the C-alpha coordinates, the
residue pairs of the program's printed alignment, and the numbers the reference's pipeline parses from its output
("RMSD of the common residues", "TM-score").  Output: tests/golden/align_tmscore.json (data only).
usage: python tests/golden/make_golden_align.py"""
import json
import os
import random
import re
import shutil
import subprocess
import tempfile

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "align_tmscore.json")
THREE = {'A': 'ALA', 'R': 'ARG', 'N': 'ASN', 'D': 'ASP', 'C': 'CYS', 'Q': 'GLN', 'E': 'GLU', 'G': 'GLY', 'H': 'HIS', 'I': 'ILE', 'L': 'LEU', 'K': 'LYS',
         'M': 'MET', 'F': 'PHE', 'P': 'PRO', 'S': 'SER', 'T': 'THR', 'W': 'TRP', 'Y': 'TYR', 'V': 'VAL'}
ONE = {v: k for k, v in THREE.items()}


def residues(path):
    res = {}
    for l in open(path):
        if l.startswith("ATOM"):
            res.setdefault(int(l[22:26]), []).append(l)
    return res


def write(path, res, keys, mut, first):
    with open(path, "w") as f:
        for n, k in enumerate(keys):
            for l in res[k]:
                f.write(l[:17] + mut.get(k, l[17:20]) + l[20:22] + "%4d" % (first + n) + l[26:])
        f.write("END\n")


def main():
    work = tempfile.mkdtemp()
    exe = os.path.join(work, "TMscore")
    shutil.copy(os.path.join(REF, "bin", "TMscore"), exe)
    os.chmod(exe, 0o755)
    nat = {n: residues(os.path.join(REF, "example", n + ".pdb")) for n in ("apo", "holo")}
    rng = random.Random(20251004)
    cases = []
    for t in range(48):
        na, nb = ("apo", "apo") if t % 3 == 0 else (("apo", "holo") if t % 3 == 1 else ("holo", "apo"))
        keys = sorted(nat[na])
        ka, kb = list(keys), list(keys)
        for ks in (ka, kb):
            for _ in range(rng.randint(0, 4)):                       # internal deletions
                s, n = rng.randint(0, len(ks) - 6), rng.randint(1, 9)
                del ks[s:s + n]
            if rng.random() < 0.4:                                    # terminal truncations
                del ks[:rng.randint(1, 12)]
            if rng.random() < 0.4:
                del ks[-rng.randint(1, 12):]
        mut = {k: rng.choice(list(THREE.values())) for k in rng.sample(keys, rng.randint(0, 30))}
        pa, pb = os.path.join(work, "a.pdb"), os.path.join(work, "b.pdb")
        write(pa, nat[na], ka, {}, rng.randint(1, 40))                # numbering unrelated between the files: only -seq can match them
        write(pb, nat[nb], kb, mut, rng.randint(1, 40))
        out = subprocess.run([exe, pa, pb, "-seq"], capture_output=True, text=True).stdout
        L = out.splitlines()
        i = [k for k, l in enumerate(L) if l.startswith('(":" denotes')][0]
        al1, al2 = L[i + 1], L[i + 3]
        pairs, x, y = [], 0, 0
        for c1, c2 in zip(al1, al2):
            if c1 != "-" and c2 != "-":
                pairs.append([x, y])
            x += c1 != "-"
            y += c2 != "-"
        ca = lambda res, ks: [[float(l[30:38]), float(l[38:46]), float(l[46:54])] for k in ks for l in res[k] if l[12:16].strip() == "CA"]
        cases.append(dict(seq_a="".join(ONE[nat[na][k][0][17:20]] for k in ka), seq_b="".join(ONE[mut.get(k, nat[nb][k][0][17:20])] for k in kb),
                          ca_a=ca(nat[na], ka), ca_b=ca(nat[nb], kb), pairs=pairs,
                          n_common=int(re.search(r"Number of residues in common=\s*(\d+)", out).group(1)),
                          rmsd=float(re.search(r"RMSD of  the common residues=\s*([\d.]+)", out).group(1)),
                          tm=float(re.search(r"TM-score\s*=\s*([\d.]+)", out).group(1))))
        assert len(pairs) == cases[-1]["n_common"]
    json.dump(dict(source="bin/TMscore -seq of the reference (prebuilt ELF, no source in the tree), run in the build container by this script",
                   cases=cases), open(OUT, "w"))
    shutil.rmtree(work)
    print(len(cases), "cases ->", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
