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


def _run(w, max_iter, sep_lo, sep_hi, precheck=0, skip_to=0, cartesian=0, pair_filter=0):
    return dict(w=list(w), max_iter=max_iter, sep_lo=sep_lo, sep_hi=sep_hi, precheck=precheck, skip_to=skip_to,
                cartesian=cartesian, pair_filter=pair_filter)


def _declash(runs, w, max_iter, sep_lo, sep_hi, pair_filter=0):
    """remove_clash(sf_vdw, mover, pose): every round is guarded by the rama+vdw < 10 test."""
    end = len(runs) + N_DECLASH
    for _ in range(N_DECLASH):
        runs.append(_run(w, max_iter, sep_lo, sep_hi, precheck=1, skip_to=end, pair_filter=pair_filter))


CART_MAX_L = 512  # the Cartesian step kernel handles one residue per thread, up to 512 threads (csrc/kernel_step.h, cart_body)


def build_runs(L, mode=2, cartesian_stage=None):
    """Run list for `-m mode` (folding/utils_ros/arguments.py:12).  Mode 3 (folding.py:173-186) first loads the restraints of
    the ORDERED pairs only (add_idr_rst with 1 - idr: runs with pair_filter = 1), then all of them; it needs a map set with its
    idr mask (Context.set_map(idr=...)).

    cartesian_stage: run min_mover_cart (folding.py:100-102,170) in Cartesian space, as the reference does.  None = yes for
    chains the Cartesian kernel supports (L <= 512); longer chains run that stage in torsion space with sf_cart's
    weights (no bonded term), which is what every chain did before the kernel existed."""
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
    return runs
