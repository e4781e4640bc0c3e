"""Staged minimisation protocol of the reference fold script, as a flat list of minimiser runs.

Mirrors /root/reference/folding/folding.py:74-104 (four score functions, four MinMovers, RepeatMover 3),
:118-119 (random start + declash), :125-186 (modes 0-3) and folding/utils_ros/utils_ros.py:699-703
(remove_clash: at most 5 x { if sf_vdw(pose) < 10: break; mover.apply(pose) }).

Weight vector layout (include/trx2_model.h): [atom_pair, dihedral, angle, vdw, rama, omega, cart_bonded, hbond].
Weights are the reference's folding/data/*.wts files.  The last slot carries cen_hb (5.0 in scorefxn.wts / scorefxn1.wts) in
the torsion-space score functions and hbond_sr_bb = hbond_lr_bb (3.0 each in scorefxn_cart.wts; equal, so one weight) in the
Cartesian one, all applied to ONE backbone hydrogen-bond surrogate (trx2_model.h TRX2_HB_*): Rosetta's potentials are not in
the reference tree.  Measured effect on the outcome: none within sampling noise.
"""

# folding/data/scorefxn.wts
SF = [5.0, 4.0, 4.0, 1.0, 1.0, 0.5, 0.0, 5.0]
# folding/data/scorefxn1.wts
SF1 = [3.0, 1.0, 1.0, 3.0, 1.0, 0.5, 0.0, 5.0]
# folding/data/scorefxn_vdw.wts
SF_VDW = [0.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0, 0.0]
# folding/data/scorefxn_cart.wts
SF_CART = [5.0, 4.0, 4.0, 0.5, 1.0, 0.5, 0.1, 3.0]

MAX_ITER = 1000      # folding.py:92,95,101
MAX_ITER_VDW = 500   # folding.py:98
N_REPEAT = 3         # folding.py:104
N_DECLASH = 5        # utils_ros.py:700
MAX_RUNS = 64        # include/trx2_model.h TRX2_MAX_RUNS (the protocol table lives in the step kernel's LDS)


# warm: the run may take its first step with the Hessian scale of the decoy's last stored correction pair (include/trx2_model.h
# TRX2_RUN_WARM) instead of a unit-length steepest-descent step: every run but the closing one of the relax stage (-8 % evaluations at
# unchanged outcome on 2 x 2048 decoys; profiles/r05_warm_sweep.txt).
def _run(w, max_iter, sep_lo, sep_hi, precheck=0, skip_to=0, cartesian=0, pair_filter=0, tol=0.0, warm=1):
    return dict(w=list(w), max_iter=max_iter, sep_lo=sep_lo, sep_hi=sep_hi, precheck=precheck, skip_to=skip_to,
                cartesian=cartesian, pair_filter=pair_filter, tol=tol, warm=warm)


def _declash(runs, w, max_iter, sep_lo, sep_hi, pair_filter=0):
    """remove_clash(sf_vdw, mover, pose): every round is guarded by the rama+vdw < 10 test."""
    end = len(runs) + N_DECLASH
    for _ in range(N_DECLASH):
        runs.append(_run(w, max_iter, sep_lo, sep_hi, precheck=1, skip_to=end, pair_filter=pair_filter))


CART_MAX_L = 512  # the Cartesian step kernel handles one residue per thread, up to 512 threads (csrc/kernel_step.h, cart_body)

# ---- backbone-visible part of the full-atom refinement (folding.py:200-268; include/trx2_model.h "a11-lite") ----------------
# scorefxn_fa = ref2015_cart + atom_pair 5 / dihedral 1 / angle 1, pro_close 0 (folding.py:202-205).  ref2015_cart's own terms
# live in the Rosetta database, not in the tree; what this build has of them are the backbone surrogates of the centroid stage,
# weighted here with ref2015_cart's published weights for the terms they stand in for [Rosetta, from memory -- unverified]:
# fa_rep 0.55 -> vdw (times the script's ramp), rama_prepro 0.45 -> rama, omega 0.4, cart_bonded 0.5, hbond_sr_bb = hbond_lr_bb 1.0.
SF_FA = [5.0, 1.0, 1.0, 0.55, 0.45, 0.4, 0.5, 1.0]
# Scale of the omega / bonded terms in the relax stage.  SF_FA's weights were written for Rosetta's terms.  omega: since round 5 the term IS
# fitted to ref2015's column of the reference decoys' energy tables (include/trx2_model.h TRX2_OMEGA_FIT), so its scale is 1 (rounds 1-4:
# 0.4 on a tether 13 x Rosetta's).  cart_bonded: the surrogate is 25 x Rosetta's column (tests/test_pose_energies.py); 0.4 is as far
# towards Rosetta's scale as the outcome improved (round 4, 2 x 1024 decoys per map).  TRX2_SF_FA_SCALE="omega,bonded" overrides (model scans).
import os as _os


def _sf_fa_scale():
    txt = _os.environ.get("TRX2_SF_FA_SCALE", "1.0,0.4")
    try:
        v = tuple(float(x) for x in txt.split(","))
        if len(v) != 2 or not all(0.0 <= x <= 100.0 for x in v):
            raise ValueError
    except ValueError:
        raise ValueError(f"TRX2_SF_FA_SCALE must be two non-negative numbers 'omega,bonded', got {txt!r}") from None
    return v


SF_FA_SCALE = _sf_fa_scale()
# folding/data/1relax_round1.txt, 2relax_round2.txt: `ramp_repack_min <fa_rep scale> <min tolerance> <coord-cst weight> <max iter>`
RELAX_RAMP_TORSION = [(0.02, 0.01, 100), (0.25, 0.01, 100), (0.55, 0.01, 100), (1.0, 0.00001, 100)]
RELAX_RAMP_CART = [(0.02, 0.01, 50), (0.25, 0.01, 50), (0.55, 0.01, 100), (1.0, 0.00001, 200)]
# The reference hands MinMover 1e-4 and this minimiser needs 1e-6 for the same outcome (include/trx2_model.h TRX2_MIN_TOL,
# DESIGN.md deviation 1): the scripts' tolerances are scaled by the same factor 0.01.
RELAX_TOL_SCALE = 0.01
FILTER_RELAX1, FILTER_RELAX2 = 2, 3   # include/trx2_model.h TRX2_FILTER_RELAX*


def relax_runs(L, cartesian_stage=True):
    """The two FastRelax rounds as minimiser runs on the backbone (no side chains to repack): round 1 = restraints re-selected at
    PCUT 0.15 without glycine pairs, torsion ramps x 2 then Cartesian ramps x 1 (1relax_round1.txt); round 2 = PCUT 0.30,
    Cartesian ramps x 2 (2relax_round2.txt); then the closing Cartesian minimisation without restraints (folding.py:257-263:
    ref2015_cart, tolerance 1e-5, 100 iterations).  `accept_to_best` and the idealize step are not replicated."""
    runs = []

    def ramps(table, cart, flt, repeat):
        for _ in range(repeat):
            for scale, tol, it in table:
                w = list(SF_FA)
                w[3] = SF_FA[3] * scale
                w[5], w[6] = SF_FA[5] * SF_FA_SCALE[0], SF_FA[6] * SF_FA_SCALE[1]
                if not cart:
                    w[6] = 0.0
                runs.append(_run(w, it, 1, L, cartesian=1 if cart else 0, pair_filter=flt, tol=tol * RELAX_TOL_SCALE))

    ramps(RELAX_RAMP_TORSION, False, FILTER_RELAX1, 2)
    ramps(RELAX_RAMP_CART, cartesian_stage, FILTER_RELAX1, 1)
    ramps(RELAX_RAMP_CART, cartesian_stage, FILTER_RELAX2, 2)
    w = [0.0, 0.0, 0.0] + SF_FA[3:]
    w[5], w[6] = SF_FA[5] * SF_FA_SCALE[0], SF_FA[6] * SF_FA_SCALE[1]
    # The closing run starts COLD (a unit-length first step): that kick and the 100 iterations behind it are what untwists the peptides the
    # restrained stages leave (2 x 2048 decoys, profiles/r05_warm_sweep.txt: twisted beyond 60 degrees 0.1 / 0.3 % cold, 1.9 / 6.0 % warm).
    runs.append(_run(w, 100, 1, L, cartesian=1 if cartesian_stage else 0, pair_filter=FILTER_RELAX2, tol=0.00001 * RELAX_TOL_SCALE, warm=0))
    return runs


def build_runs(L, mode=2, cartesian_stage=None, fastrelax=False):
    """Run list for `-m mode` (folding/utils_ros/arguments.py:12).  Mode 3 (folding.py:173-186) first loads the restraints of
    the ORDERED pairs only (add_idr_rst with 1 - idr: runs with pair_filter = 1), then all of them; it needs a map set with its
    idr mask (Context.set_map(idr=...)).

    cartesian_stage: run min_mover_cart (folding.py:100-102,170) in Cartesian space, as the reference does.  None = yes for
    chains the Cartesian kernel supports (L <= 512); longer chains run that stage in torsion space with sf_cart's
    weights (no bonded term), which is what every chain did before the kernel existed.

    fastrelax: append the backbone-visible part of the reference's full-atom refinement (relax_runs)."""
    if cartesian_stage is None:
        cartesian_stage = L <= CART_MAX_L
    runs = []
    # folding.py:119  remove_clash(sf_vdw, min_mover_vdw, pose) -- no restraints loaded yet
    _declash(runs, SF_VDW, MAX_ITER_VDW, 0, 0)
    if mode == 0:
        stages = [(1, 12), (1, 24), (1, L)]      # add_rst accumulates: folding.py:129,136,143
    elif mode == 1:
        stages = [(3, 24), (3, L)]               # folding.py:152,159
    elif mode == 2:
        stages = [(1, L)]                        # folding.py:168
    elif mode == 3:
        stages = [(1, L, 1), (1, L, 0)]          # folding.py:177,183: add_idr_rst has no separation window; restraints accumulate
    else:
        raise ValueError(f"unknown mode {mode}")
    for st in stages:
        lo, hi, flt = st if len(st) == 3 else (st[0], st[1], 0)
        for _ in range(N_REPEAT):                # repeat_mover.apply
            runs.append(_run(SF, MAX_ITER, lo, hi, pair_filter=flt))
        # min_mover_cart.apply: Cartesian-space L-BFGS on sf_cart
        w_cart = SF_CART if cartesian_stage else SF_CART[:6] + [0.0, SF_CART[7]]  # no bonded term in torsion space
        runs.append(_run(w_cart, MAX_ITER, lo, hi, cartesian=1 if cartesian_stage else 0, pair_filter=flt))
        _declash(runs, SF1, MAX_ITER, lo, hi, pair_filter=flt)    # remove_clash(sf_vdw, min_mover1, pose)
    if fastrelax:
        runs += relax_runs(L, cartesian_stage)
    if len(runs) > MAX_RUNS:   # mode 0 with the relax stage is the longest: 32 + 21 = 53
        raise ValueError(f"protocol of {len(runs)} runs exceeds TRX2_MAX_RUNS = {MAX_RUNS} (include/trx2_model.h)")
    return runs
