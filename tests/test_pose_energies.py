"""CPU: the surrogate backbone terms against the only energy-level numbers of Rosetta's that the reference tree holds -- the
per-residue POSE_ENERGIES_TABLE of its eight committed decoys (ref2015_cart, weighted; tests/golden/pose_energies.json, made by
make_golden_pose_energies.py from example/output/seq/pred_pdb/conf_*.pdb).

What this can and cannot pin.  Those tables are FULL-ATOM energies of the refined decoys; the fold's model is a backbone with
surrogates for vdw / rama / omega / cart_bonded / hydrogen bonds (include/trx2_model.h), evaluated here on the same backbone
coordinates.  Checked: (i) the table's weights ARE the ref2015_cart weights protocol.SF_FA uses for the relax stage (they were
"from memory" until this file existed); (ii) the provenance of every decoy (which map and stage it came from); (iii) per term, the
rank correlation over residues between surrogate and Rosetta column and the ratio of their totals.  All of it is REPORTED
(profiles/r04_pose_energies.txt, written when TRX2_WRITE_REPORT=1); asserted are measured values with margin, as floors that catch
a sign error or a mis-assigned residue, no more -- every surrogate is a different function of the same geometry than Rosetta's term
(omega: a quadratic tether at 180 degrees vs ref2015's phi/psi-dependent tether, which is negative for a third of the residues;
rama: a mixture of six basins, >= 0, vs sequence-dependent tables relative to the average; cart_bonded: backbone-only harmonic
terms vs all atoms; fa_rep: five backbone atoms vs every atom with its hydrogens).  Rosetta prints the backbone hydrogen-bond
energies per POSE only (its per-residue columns are zero: the term is context dependent), so that term is compared by totals.
Findings, with the constants of rounds 1-3: the bonded surrogate ranked residues like cart_bonded does (rho
0.6-0.75) at 25 x the energy for the same deviations; the omega tether was 13 x ref2015's; the hydrogen-bond total is half of
Rosetta's; the rama surrogate sits 1.3 per residue above Rosetta's.  Round 4 acts on it where ref2015 applies: the relax stage
scales the two surrogates by 0.4 (protocol.SF_FA_SCALE: as far towards Rosetta's scale as the outcome improves) and remove_clash's
guard gets the rama offset (trx2_model.h TRX2_RAMA_GUARD_OFFSET); the centroid stage keeps its calibration.
Round 5 FITS the rama and omega terms to these tables (tools/fit_backbone_terms.py; include/trx2_model.h TRX2_RAMA_FIT_* / TRX2_OMEGA_FIT):
rama_prepro is a two-body energy that the table splits half / half between residue i and i + 1, so the comparison convolves the model's
per-residue term the same way; rank correlations rama 0.15 -> 0.7, omega 0.32 -> 0.8 (the shipped surface is the fit shrunk to one half and
the tether 3 x as stiff about its fitted centre: what the outcome scans chose, profiles/r05_model_scan*.txt).  The constants in the header
are checked against a re-run of the fit."""
import importlib
import json
import os

import numpy as np
import pytest
from scipy.stats import spearmanr

from oracle import oracle as O

P = importlib.import_module("trrosettax2-dynamics_amd").protocol
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROVENANCE = {"conf_1_1": "Xray/initial0", "conf_1_2": "Xray/initial1", "conf_1_3": "NMR/seq1", "conf_1_4": "NMR/seq2",
              "conf_2_1": "NMR/initial0", "conf_2_2": "NMR/initial1", "conf_2_3": "Xray/seq3", "conf_2_4": "Xray/seq4"}


@pytest.fixture(scope="module")
def table(golden_dir):
    return json.load(open(os.path.join(golden_dir, "pose_energies.json")))


def test_weights_and_provenance(table):
    for name, want in PROVENANCE.items():                       # which file became which conf_*
        assert table[name]["provenance"].endswith(f"pred_pdb/{want}.pdb"), (name, table[name]["provenance"])
    w = table["conf_2_1"]["weights"]
    # protocol.SF_FA = [atom_pair, dihedral, angle, vdw <- fa_rep, rama <- rama_prepro, omega, cart_bonded, hbond <- hbond_sr_bb = hbond_lr_bb]
    assert [w["fa_rep"], w["rama_prepro"], w["omega"], w["cart_bonded"], w["hbond_sr_bb"]] == P.SF_FA[3:8] and w["hbond_lr_bb"] == w["hbond_sr_bb"]
    assert all(t["weights"] == w for t in table.values())


def surrogate_and_rosetta(table, golden_dir, seq):
    ref = np.load(os.path.join(golden_dir, "ref_decoys.npz"))
    m = np.load(os.path.join(golden_dir, "seq_NMR.npz"))
    tab = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)      # (the split needs the sequence only: proline donors)
    rows = {}
    for name in sorted(PROVENANCE):
        x = np.asarray(ref[name], np.float64).copy()
        b, c = x[:, 1] - x[:, 0], x[:, 2] - x[:, 1]
        bad = ~np.isfinite(x[:, 4]).all(1)                      # glycine: the model's own virtual C-beta
        x[bad, 4] = (-0.58273431 * np.cross(b, c) + 0.56802827 * b - 0.54067466 * c + x[:, 1])[bad]
        sur = O.per_residue_terms(tab, x)
        # the split adds up to the model's totals
        e, _ = O.energy_cart(tab, x, [0, 0, 0, 1, 0, 0, 0, 1])
        assert abs(sur[:, 2].sum() - e[8]) < 1e-9 * max(1, abs(e[8])) and abs(sur[:, 4].sum() - e[4]) < 1e-9 * max(1, abs(e[4]))
        _, et, _ = O.eval_cart(tab, x, [0, 0, 0, 0, 1, 1, 1, 0], grad=False)
        assert abs(sur[:, 0].sum() - et[6]) < 1e-9 * max(1, et[6]) and abs(sur[:, 1].sum() - et[5]) < 1e-9 * max(1, abs(et[5])) and abs(sur[:, 3].sum() - et[7]) < 1e-9 * max(1, et[7])
        pr = table[name]["per_residue"]
        ros = np.stack([pr["omega"], pr["rama_prepro"], np.add(pr["hbond_sr_bb"], pr["hbond_lr_bb"]), pr["cart_bonded"], pr["fa_rep"]], 1)
        assert np.all(ros[:, 2] == 0.0)                         # Rosetta prints backbone hydrogen bonds per pose only
        rows[name] = (sur, ros, table[name]["pose"]["hbond_sr_bb"] + table[name]["pose"]["hbond_lr_bb"])
    return rows


def test_surrogate_terms_against_the_reference_decoys_energy_tables(table, golden_dir, seq):
    rows = surrogate_and_rosetta(table, golden_dir, seq)
    names = ("omega", "rama", "hbond_bb", "cart_bonded", "repulsion")
    w = [P.SF_FA[5], P.SF_FA[4], P.SF_FA[7], P.SF_FA[6], P.SF_FA[3]]      # the table holds WEIGHTED energies: weigh the surrogates alike
    for dec in rows:          # rama_prepro is two-body in Rosetta, split half / half between i and i + 1 (tools/fit_backbone_terms.py): ours alike
        sur = rows[dec][0]; r = sur[:, 1].copy(); sur[:, 1] = 0.5 * (r + np.r_[0.0, r[:-1]])
    lines = ["surrogate backbone terms (include/trx2_model.h, weighted with ref2015_cart's weights) vs the per-residue columns of the reference",
             "decoys' POSE_ENERGIES_TABLE; per decoy: Spearman rank correlation over the 90 residues | total surrogate / total Rosetta",
             "(hbond_bb: Rosetta prints hbond_sr_bb + hbond_lr_bb per pose only -- no per-residue ranking; repulsion: five backbone atoms against fa_rep over all atoms)", ""]
    lines.append("%-9s" % "decoy" + "".join("%26s" % n for n in names))
    rho = {n: [] for n in names}
    ratio = {n: [] for n in names}
    for dec, (sur, ros, hb_pose) in rows.items():
        cells = []
        for k, n in enumerate(names):
            a, b = w[k] * sur[:, k], ros[:, k]
            bsum = hb_pose if n == "hbond_bb" else b.sum()
            r = spearmanr(a, b).statistic if (a.std() > 0 and b.std() > 0) else float("nan")
            rho[n].append(r); ratio[n].append(a.sum() / bsum if abs(bsum) > 1e-9 else float("nan"))
            cells.append(("%10.2f" % r if np.isfinite(r) else "         -") + " | %6.2f / %7.2f" % (a.sum(), bsum))
        lines.append("%-9s" % dec + "".join("%26s" % c for c in cells))
    med = lambda v: float(np.median([x for x in v if np.isfinite(x)])) if any(np.isfinite(x) for x in v) else float("nan")
    lines.append("%-9s" % "median" + "".join("%10.2f | ratio %9.2f" % (med(rho[n]), med(ratio[n])) for n in names))
    report = "\n".join(lines)
    print("\n" + report)
    if os.environ.get("TRX2_WRITE_REPORT") == "1":
        open(os.path.join(ROOT, "profiles", "r06_pose_energies.txt"), "w").write(report + "\n")
    # measured medians over the eight decoys (round 5, fitted rama / omega): rho omega 0.80, rama 0.70, cart_bonded 0.70, repulsion 0.37;
    # totals: hydrogen bonds 0.51 x Rosetta's, cart_bonded 25 x; omega and rama carry Rosetta's level where they were fitted (the omega
    # total is a small difference of positive and negative residues: no ratio asserted).  VERDICT r4 item 4 asked for >= 0.5 on rama and omega.
    # Round 6: the helix term is a constant per CLASS (no per-residue-type constants fitted on one sequence in the default model) --
    # rama 0.70 -> 0.51 (min 0.36) on these eight decoys, the price of not fitting 16 residue types on them; the outcome did not move
    # (profiles/r06_fit_generalisation.txt).  Rounds 1-4's six-basin prior alone: 0.15.
    assert np.nanmedian(rho["omega"]) >= 0.65 and np.nanmedian(rho["rama"]) >= 0.45 and min(rho["omega"]) >= 0.5 and min(rho["rama"]) >= 0.3, rho
    assert np.nanmedian(rho["cart_bonded"]) > 0.5 and np.nanmedian(rho["repulsion"]) > 0.2, rho
    assert 0.35 < np.nanmedian(ratio["hbond_bb"]) < 0.75 and 12 < np.nanmedian(ratio["cart_bonded"]) < 50, ratio
    tot_s = np.array([w[1] * rows[d][0][:, 1].sum() for d in rows]); tot_r = np.array([rows[d][1][:, 1].sum() for d in rows])
    assert np.abs(tot_s - tot_r).max() < 8.0, (tot_s, tot_r)      # rama totals within 8 units of Rosetta's on every decoy (measured < 5; rounds 1-4: 100 apart)


def test_header_constants_are_the_fit(golden_dir):
    """include/trx2_model.h's TRX2_RAMA_FIT_* / TRX2_OMEGA_FIT lines are the output of tools/fit_backbone_terms.py on the committed fixtures."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fit_backbone_terms", os.path.join(ROOT, "tools", "fit_backbone_terms.py"))
    mod = importlib.util.module_from_spec(spec)
    import sys
    argv, sys.argv = sys.argv, ["fit_backbone_terms.py", ROOT]
    try:
        spec.loader.exec_module(mod)
        lines = mod.main(quiet=True)
    finally:
        sys.argv = argv
    hdr = open(os.path.join(ROOT, "include", "trx2_model.h")).read()
    for ln in lines:
        assert ln in hdr, ln
