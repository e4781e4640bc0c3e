"""PDB text in and out for the fold path.

write_pdb replaces pose.dump_pdb (/root/reference/folding/folding.py:273) for backbone decoys: ATOM records N, CA, C, O,
CB (no CB for Gly -- folding.py:190-193 mutates Ala back to Gly before the dump), chain A, residues 1..L.  That is all
the downstream readers need (utils_trX2dy/utils.py:252-291 reads N/CA/C/CB + residue names; PPBuilder needs C-N < 1.8 A;
TMscore needs CA).  Files are written to a temporary name and renamed, so a partial PDB is never left behind.

read_backbone replaces the Biopython parse of get_atom_positions_pdb (utils.py:252-291, retain_all_res=False): first
model, amino-acid ATOM records, one row per residue in file order, NaN for absent atoms.  It reads both this package's
output and the reference's full-atom PyRosetta PDBs.
"""
import functools
import os
import threading

import numpy as np

AA3 = {"A": "ALA", "R": "ARG", "N": "ASN", "D": "ASP", "C": "CYS", "Q": "GLN", "E": "GLU", "G": "GLY", "H": "HIS",
       "I": "ILE", "L": "LEU", "K": "LYS", "M": "MET", "F": "PHE", "P": "PRO", "S": "SER", "T": "THR", "W": "TRP",
       "Y": "TYR", "V": "VAL", "U": "SEC", "X": "UNK"}
# utils.py:25-54 (res_name_dict), three-letter -> one-letter incl. the modified residues it lists
AA1 = {v: k for k, v in AA3.items()}
AA1.update({"PHD": "D", "MSE": "M", "ASX": "B", "GLX": "Z", "XLE": "J", "XAA": "X"})
ATOMS = ("N", "CA", "C", "O", "CB")
ELEMENT = {"N": "N", "CA": "C", "C": "C", "O": "O", "CB": "C"}


def read_fasta(path):
    """first chain of a FASTA file (folding.py:17-29)"""
    seq = ""
    with open(path) as f:
        for line in f:
            if line.startswith(">"):
                if seq:
                    break
                continue
            seq += line.rstrip()
    return seq


@functools.lru_cache(maxsize=64)
def _atom_template(seq):
    """per sequence: the constant text around every ATOM record's coordinates (a chain of single-decoy folds writes the same 5 L records
    hundreds of times; formatting them field by field was 4.5 ms per decoy at L=150, 4 % of an iteration of run_inference's loop)"""
    rows, serial = [], 1
    for i, aa in enumerate(seq):
        res = AA3.get(aa.upper(), "UNK")
        for k, name in enumerate(ATOMS):
            if name == "CB" and res == "GLY":
                continue
            aname = f" {name:<3s}"  # element right-justified in cols 13-14 for one-letter elements
            rows.append((i, k, f"ATOM  {serial:5d} {aname} {res} A{i + 1:4d}    ", f"  1.00  0.00          {ELEMENT[name]:>2s}"))
            serial += 1
    tail = [f"TER   {serial:5d}      {AA3.get(seq[-1].upper(), 'UNK')} A{len(seq):4d}", "END"]
    return tuple(rows), tuple(tail)


def write_pdb(path, seq, xyz, remarks=()):
    xyz = np.asarray(xyz, dtype=np.float64)
    L = len(seq)
    if xyz.shape != (L, 5, 3):
        raise ValueError(f"xyz must be ({L}, 5, 3), got {xyz.shape}")
    if not np.all(np.isfinite(xyz)):
        raise ValueError("refusing to write a PDB with non-finite coordinates")
    rows, tail = _atom_template(seq)
    c = xyz.tolist()      # Python floats: "%8.3f" of them is the same text as the f-string of the numpy scalars, several times faster
    lines = [f"REMARK   {r}" for r in remarks]
    lines += [pre + "%8.3f%8.3f%8.3f" % tuple(c[i][k]) + post for i, k, pre, post in rows]
    lines += tail
    tmp = f"{path}.{os.getpid()}.{threading.get_ident()}.tmp"
    with open(tmp, "w") as f:
        f.write("\n".join(lines) + "\n")
    os.replace(tmp, path)


def as_read_from_pdb(seq, xyz):
    """What read_backbone(path) returns for a file write_pdb(path, seq, xyz) has written, without the file: coordinates after
    the "%8.3f" round trip (float32 -> exact double -> nearest multiple of 0.001, ties to even as the formatter does ->
    nearest double -> float32; x * 1000 is exact in double for a float32 x) and NaN for glycine's absent CB.  Standard
    residues only (the writer's UNK records are skipped by the reader)."""
    if any(a.upper() not in AA3 for a in seq):
        raise ValueError("non-standard residue: parse the file instead")
    v = np.rint(np.asarray(xyz, np.float32).astype(np.float64) * 1000.0) / 1000.0
    out = v.astype(np.float32)
    out[np.array([a.upper() == "G" for a in seq]), ATOMS.index("CB")] = np.nan
    return out, "".join(a.upper() for a in seq)


def read_backbone(path):
    """-> (xyz[L,5,3] float32 with NaN for absent atoms, one-letter sequence)"""
    rows, names, index = [], [], {}
    with open(path) as f:
        for line in f:
            if line.startswith("ENDMDL"):
                break
            if not line.startswith("ATOM"):
                continue
            if line[16] not in (" ", "A"):  # first alternate location only
                continue
            res = line[17:20].strip()
            if res not in AA1:
                continue
            key = (line[21], line[22:27])
            if key not in index:
                index[key] = len(rows)
                rows.append(np.full((5, 3), np.nan, np.float32))
                names.append(res)
            an = line[12:16].strip()
            if an in ATOMS:
                rows[index[key]][ATOMS.index(an)] = (float(line[30:38]), float(line[38:46]), float(line[46:54]))
    if not rows:
        raise ValueError(f"{path}: no amino-acid ATOM records")
    return np.stack(rows), "".join(AA1[n] for n in names)
