/* trx2_model.h -- constants that DEFINE the fold's energy model and protocol.
 *
 * Shared (as data, not code) by the HIP product path (trrosettax2-dynamics_amd/csrc) and by the CPU
 * oracle (oracle/trx2_oracle.c) so that both evaluate the same function.  Plain C, no dependencies.
 *
 * Provenance of every number:
 *  - restraint-table parameters: /root/reference/folding/data/params.json:1-16 and
 *    folding/utils_ros/utils_ros.py:18-31,55-61,82,125 (gen_rst)
 *  - score-function weights: folding/data/scorefxn{,1,_vdw,_cart}.wts (quoted in SURVEY.md 3.2)
 *  - selection thresholds: folding/utils_ros/utils_ros.py:719-723 (add_rst)
 *  - random start table: folding/utils_ros/utils_ros.py:667-696 (random_dihedral)
 *  - minimiser settings: folding/folding.py:91-104 (tolerance 1e-4, max_iter 1000/500, RepeatMover 3)
 *  - ideal backbone geometry and soft-sphere radii: measured on the reference's 8 committed PyRosetta
 *    decoys (example/output/seq/pred_pdb/conf_*.pdb) by tools/derive_constants.py, because the Rosetta
 *    database (ICOOR ideals, AtomVDW, Rama tables) is not in the reference tree.
 */
#ifndef TRX2_MODEL_H
#define TRX2_MODEL_H

/* ---- atoms per residue, in storage order ------------------------------------------------------- */
#define TRX2_NATOM 5 /* N, CA, C, O, CB */
#define TRX2_AT_N 0
#define TRX2_AT_CA 1
#define TRX2_AT_C 2
#define TRX2_AT_O 3
#define TRX2_AT_CB 4
#define TRX2_RES_STRIDE 16 /* floats per residue in device buffers: 5 atoms x 3 + 1 pad = 64 B */

/* ---- distogram / table geometry (reference bins) -------------------------------- */
#define TRX2_ND_BINS 37 /* dist  npz last dim */
#define TRX2_NO_BINS 25 /* omega, theta */
#define TRX2_NP_BINS 13 /* phi */
#define TRX2_KD 35      /* knots of the distance spline: 3 repulsive + 32 contact bins (utils_ros.py:60-64) */
#define TRX2_KD_AF2 60  /* ... of gen_rst_af2's C-alpha tables: 3 repulsive + 57 AlphaFold bin edges (utils_ros.py:174-181) */
#define TRX2_KO 28      /* knots of omega / theta splines: 24 + 4 wrap-pad (utils_ros.py:81-87)          */
#define TRX2_KP 16      /* knots of phi spline: 12 + 4 mirror-pad (utils_ros.py:124-130)                */
#define TRX2_KTOT (TRX2_KD + TRX2_KO + TRX2_KO + TRX2_KP) /* 107 knots per ordered pair */
#define TRX2_KTOT_MAX (TRX2_KD_AF2 + TRX2_KO + TRX2_KO + TRX2_KP)
#define TRX2_GEN_PCUT 0.05 /* utils_ros.py:18 (literal, ignores -pd) */

/* selection mask bits per ORDERED pair (a,b) */
#define TRX2_M_DIST 1
#define TRX2_M_OMEGA 2
#define TRX2_M_THETA 4
#define TRX2_M_PHI 8

/* ---- ideal backbone geometry (tools/derive_constants.py on the reference decoys) --------------- */
#define TRX2_B_N_CA 1.458
#define TRX2_B_CA_C 1.524
#define TRX2_B_C_N 1.334
#define TRX2_B_C_O 1.232
#define TRX2_A_N_CA_C 111.4  /* degrees */
#define TRX2_A_CA_C_N 117.0
#define TRX2_A_C_N_CA 121.0
#define TRX2_A_CA_C_O 120.4
/* CB = CA + KA*(b x c) + KB*b + KC*c with b = CA-N, c = C-CA (same construction as the reference's
 * virtual CB, utils_trX2dy/utils.py:132-135; coefficients refitted to the decoys' real CB) */
#define TRX2_CB_KA (-0.58433326)
#define TRX2_CB_KB (0.57201293)
#define TRX2_CB_KC (-0.53795593)

/* ---- energy terms ------------------------------------------------------------------------------- */
enum {
  TRX2_E_DIST = 0, /* atom_pair_constraint : CB-CB spline               */
  TRX2_E_OMEGA,    /* dihedral_constraint  : CA CB CB CA                */
  TRX2_E_THETA,    /* dihedral_constraint  : N CA CB CB                 */
  TRX2_E_PHI,      /* angle_constraint     : CA CB CB                   */
  TRX2_E_VDW,      /* soft-sphere repulsion (surrogate of centroid vdw) */
  TRX2_E_RAMA,     /* -ln mixture over the reference's 6 basins         */
  TRX2_E_OMEGA_BB, /* peptide-bond planarity tether (Rosetta "omega")   */
  TRX2_E_CART,     /* harmonic bonded geometry (Cartesian stage only)   */
  TRX2_E_HB,       /* backbone hydrogen bonds (surrogate of cen_hb / hbond_sr_bb + hbond_lr_bb) */
  TRX2_NTERMS
};
/* weight vector layout: w[0]=atom_pair, w[1]=dihedral (omega & theta), w[2]=angle (phi), w[3]=vdw,
 * w[4]=rama, w[5]=omega_bb, w[6]=cart_bonded, w[7]=backbone hydrogen bonds */
#define TRX2_NW 8

/* soft-sphere: E = VDW_SCALE * sum_{|i-j|>=VDW_MINSEP} max(0, r0^2 - d^2)^2 / r0^2 */
#define TRX2_VDW_SCALE 0.8
#define TRX2_VDW_MINSEP 3
/* skip residue pairs with |CA-CA|^2 above this.  An atom pair contributes only when d < r0, and d >= |CA-CA| - |CA-X_a| -
 * |CA-X_b|; with the ideal extents (N 1.458, C 1.524, CB 1.53, O 2.40) the largest r0 + extents is O-O: 3.01 + 4.80 = 7.81
 * (next: C-O 7.30, CB-CB 7.21), and a hydrogen bond needs |CA-CA| <= 1.458 + 1.01 + 3.00 + 2.40 = 7.87 (N-H...O within its
 * support).  8.0 leaves 0.13-0.19 A for the Cartesian stage's geometry, whose bonded term holds bonds to 0.01 A (measured sd).
 * (8.5 until round 3: a sixth more contacts to walk, none of which can contribute.) */
#define TRX2_VDW_CUT2 (8.0 * 8.0)
/* r0 by atom-type pair, order N CA C O CB (0.1-percentile closest approach, |i-j|>=3, in the decoys) */
#define TRX2_VDW_R0_INIT                                                                               \
  {                                                                                                    \
    {4.28, 4.14, 3.49, 2.80, 3.72}, {4.14, 4.67, 4.23, 3.36, 4.02}, {3.49, 4.23, 4.42, 3.38, 3.97},    \
        {2.80, 3.36, 3.38, 3.01, 3.25}, {                                                              \
      3.72, 4.02, 3.97, 3.25, 4.15                                                                     \
    }                                                                                                  \
  }

/* backbone hydrogen bond N-H(i) ... O=C(j), |i-j| >= HB_MINSEP, donor i >= 1 and not proline.  Surrogate of the terms the
 * reference weights with cen_hb 5.0 (folding/data/scorefxn.wts:1, scorefxn1.wts:1) and hbond_sr_bb / hbond_lr_bb 3.0 each
 * (scorefxn_cart.wts:1-2; equal weights, so the |i-j| <= 4 split is not needed).  Rosetta's potentials are not in the tree:
 *   H   = N + HB_B_NH * unit( unit(N - C(i-1)) + unit(N - CA) )          in-plane bisector
 *   E   = -HB_SCALE * f_d(|H-O|) * max(0, cos(N-H..O))^2 * max(0, cos(H..O=C))^2
 *   f_d = (1 - ((d - HB_D0) / HB_R)^2)^2  inside |d - HB_D0| < HB_R, 0 outside  (C1, compact support 0.9 .. 3.0 A)
 * cos(N-H..O) = unit(H-N).unit(O-H) and cos(H..O=C) = unit(O-C).unit(H-O) are both +1 for a straight N-H...O=C.
 * The well sits where the reference's decoys put their closest N...O approaches (2.69 min / 2.83 A at the 0.1 percentile,
 * |N-O| = 1.01 + 1.95 for a straight bond). */
#define TRX2_HB_B_NH 1.01
#define TRX2_HB_D0 1.95
#define TRX2_HB_R 1.05
#define TRX2_HB_MINSEP 3
#ifndef TRX2_HB_SCALE
#define TRX2_HB_SCALE 1.0
#endif

/* rama surrogate: E_i = -ln( (sum_k p_k exp(KAPPA (cos(phi-phi_k) + cos(psi-psi_k) - 2)) + FLOOR) / P_REF )
 * for residues 2..L-1; basins = the reference's start table (utils_ros.py:667-673), degrees */
#define TRX2_RAMA_NB 6
#define TRX2_RAMA_INIT                                                                                 \
  {                                                                                                    \
    {-140.0, 153.0, 0.135}, {-72.0, 145.0, 0.155}, {-122.0, 117.0, 0.073}, {-82.0, -14.0, 0.122},      \
        {-61.0, -41.0, 0.497}, {                                                                       \
      57.0, 39.0, 0.018                                                                                \
    }                                                                                                  \
  }
#define TRX2_RAMA_KAPPA 8.0
#define TRX2_RAMA_FLOOR 1.0e-3
#define TRX2_RAMA_PREF 0.497
/* cumulative thresholds of random_dihedral (utils_ros.py:678-695): r<=t[k] picks basin k */
#define TRX2_RAND_CUM_INIT {0.135, 0.29, 0.363, 0.485, 0.982, 2.0}

/* ---- rama / omega FITTED to the only Rosetta energies the reference tree holds (round 5) ---------------------------------
 * The eight committed decoys carry ref2015_cart's per-residue energy table (tests/golden/pose_energies.json).  tools/fit_backbone_terms.py
 * recovers the per-residue rama_prepro terms from it (Rosetta splits that two-body energy half / half between residue i and i + 1; the
 * recursion closes to 1e-4 on all eight decoys) and fits, by ridge regression, leave-one-decoy-out validated:
 *   rama_i  = six-basin prior(phi, psi) + c_class + sum_k a_class,k f_k(phi, psi) + h_class r_alpha(phi, psi)
 *             f = cos psi, sin psi, cos(phi - psi), sin(phi - psi), cos phi, sin phi, cos(phi + psi), sin(phi + psi);
 *             classes general / glycine (own surface and helix constant), proline / before a proline (a constant each: 8 samples at one place);
 *             r_alpha = posterior weight of the two right-handed helical basins of the mixture, h_class a helix constant per class
 *             (round 5 fitted one per residue TYPE on this one sequence and shipped it for every protein: see TRX2_RAMA_FIT_HELIX_CLASS below)
 *   omega_i = A(psi_i) + B(psi_i) x + C(psi_i) x^2,  x = (omega_i - 180 deg) / 10 deg,  A, B, C = q0 + q1 cos psi_i + q2 sin psi_i
 *             (ref2015's tether has a conformation-dependent centre and width; psi_i carries most of it; C > 0 for every psi)
 * Rank correlation over residues with Rosetta's columns (median of the eight decoys): rama 0.21 -> 0.79 held out, omega 0.33 -> 0.81.
 * The constants below are that script's output (tests/test_pose_energies.py re-runs it and compares).  All eight decoys are folds of
 * ONE sequence: the validation says nothing about other proteins (tests/diag/fit_generalisation.py, profiles/r06_fit_generalisation.txt:
 * leave-one-chain-out on OUTCOME, and weak-restraint folds of non-helical targets with the fitted terms on and off).
 * TRX2_RAMA_FIT_ON 0 / TRX2_OMEGA_FIT_ON 0 restore rounds 1-4's terms (model scans).
 * TRX2_OMEGA_STIFF: the fitted tether's stiffness about its own (psi-dependent) centre, E = A + STIFF (B x + C x^2).  1 is ref2015's;
 * with it this model's chains twist their peptides (2 x 2048 decoys, default protocol: 1.9 % / 3.0 % of the decoys carry a peptide beyond
 * 60 degrees, --no-fastrelax 30 % / 44 %; none of the reference's eight does) -- the other terms that hold Rosetta's peptides flat are not
 * in the tree.  Calibrated by outcome like rounds 1-4's TRX2_OMEGA_K: 3 leaves 0.1 / 0.2 % under the default protocol and the best global
 * figures of the scan (profiles/r05_model_scan3.txt); --no-fastrelax keeps 7 % / 28 % (rounds 1-4's tether: 4 % / 12 %). */
#ifndef TRX2_RAMA_FIT_ON
#define TRX2_RAMA_FIT_ON 1
#endif
#ifndef TRX2_OMEGA_FIT_ON
#define TRX2_OMEGA_FIT_ON 1
#endif
#ifndef TRX2_OMEGA_STIFF
#define TRX2_OMEGA_STIFF 3.0
#endif
/* TRX2_RAMA_FIT_SHRINK: the fitted class surfaces (not the constants, not the helix term) enter scaled by this factor -- a stronger ridge
 * after the fact.  As fitted (1.0) the surface costs the NMR map's global parity (median C-alpha RMSD to the reference decoys 0.746 ->
 * 0.864 A on 2 x 2048 decoys); at one half it keeps most of the local gain (mean |dphi|, |dpsi| to the closer reference decoy 17.7 / 17.7 ->
 * 15.6 / 15.9 degrees on the NMR map, 12.8 / 11.9 -> 10.0 / 9.4 on the X-ray map) with the global figures unchanged or better (X-ray decoys
 * within 0.5 A 56 -> 65 %, mirror-trapped starts 6.9 -> 4.5 %): profiles/r05_model_scan*.txt.  A surface fitted to torsions that cluster in
 * a few basins reproduces Rosetta's ENERGIES there; its forces between the clusters are the ridge's, not Rosetta's. */
#ifndef TRX2_RAMA_FIT_SHRINK
#define TRX2_RAMA_FIT_SHRINK 0.50
#endif
/* The fit's constants (tools/fit_backbone_terms.py prints these lines; tests/test_pose_energies.py re-runs it and compares).  Each can be
 * overridden on the compiler's command line: the leave-one-chain-out builds of tests/diag/fit_generalisation.py are made that way. */
#ifndef TRX2_RAMA_FIT_GENERAL
#define TRX2_RAMA_FIT_GENERAL {-2.7974f, -1.5162f, -0.1823f, -0.4471f, 0.2686f, 3.2531f, -1.7017f, -2.6745f}
#endif
#ifndef TRX2_RAMA_FIT_GLY
#define TRX2_RAMA_FIT_GLY {-1.0537f, -0.8192f, 1.2651f, -0.2527f, 0.6021f, -1.6864f, 0.4358f, -0.5246f}
#endif
#ifndef TRX2_RAMA_FIT_CONST
#define TRX2_RAMA_FIT_CONST {0.2072f, -4.2043f, -2.2731f, -4.7437f} /* general, glycine, proline, before a proline */
#endif
/* Helix term h r_alpha(phi, psi).  Round 6: h is a constant PER CLASS (general, glycine; proline and the residue
 * before one keep their constant alone).  Round 5 shipped a propensity per residue TYPE fitted on the one sequence of the example -- 16 types
 * with 1-12 residues each, exact zeros for the four types it lacks (C, H, W, Y) -- for every protein; those per-type deviations are now a
 * diagnostic of the fit script (--per-aa) and of TRX2_RAMA_FIT_AA builds, off by default. */
#ifndef TRX2_RAMA_FIT_HELIX_CLASS
#define TRX2_RAMA_FIT_HELIX_CLASS {-0.2955f, 2.5974f, 0.0000f, 0.0000f} /* general, glycine, proline, before a proline */
#endif
#ifndef TRX2_RAMA_FIT_AA
#define TRX2_RAMA_FIT_AA 0
#endif
#ifndef TRX2_RAMA_FIT_HELIX_AA
#define TRX2_RAMA_FIT_HELIX_AA {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f} /* ACDEFGHIKLMNPQRSTVWY: deviation from the class constant (diagnostic builds) */
#endif
#ifndef TRX2_OMEGA_FIT
#define TRX2_OMEGA_FIT {-0.0435f, -0.0758f, -0.0839f, -0.2191f, -0.4417f, 0.2092f, 1.2822f, 0.2381f, -0.4129f} /* A(psi), B(psi), C(psi): q0 + q1 cos psi + q2 sin psi each; x = (omega - 180 deg) / 10 deg */
#endif
/* per-residue parameter block of the rama term, 12 floats (three float4 on the device): [0] c_class, [1..8] a_class, [9] h, [10..11] 0.
 * seq NULL = all alanine (what a map without a sequence is folded as).  Shared by the oracle and the library's host side. */
#define TRX2_RAMA_NPAR 12
static inline void trx2_rama_params(const char* seq, int i, int L, float* p) {
  static const float gen[8] = TRX2_RAMA_FIT_GENERAL, gly[8] = TRX2_RAMA_FIT_GLY, cst[4] = TRX2_RAMA_FIT_CONST, helc[4] = TRX2_RAMA_FIT_HELIX_CLASS,
                     hel[20] = TRX2_RAMA_FIT_HELIX_AA;
  static const char aa[] = "ACDEFGHIKLMNPQRSTVWY";
  const char a = seq ? seq[i] : 'A';
  const int cls = a == 'G' ? 1 : a == 'P' ? 2 : (seq && i + 1 < L && seq[i + 1] == 'P') ? 3 : 0;
  for (int k = 0; k < TRX2_RAMA_NPAR; k++) p[k] = 0.0f;
  if (!TRX2_RAMA_FIT_ON) return;
  p[0] = cst[cls];
  for (int k = 0; k < 8; k++) p[1 + k] = (float)TRX2_RAMA_FIT_SHRINK * (cls == 0 ? gen[k] : cls == 1 ? gly[k] : 0.0f);
  p[9] = helc[cls];
  if (TRX2_RAMA_FIT_AA)
    for (int k = 0; k < 20; k++) if (aa[k] == a) p[9] += hel[k];
}

/* omega_bb: E = OMEGA_K * (wrap(omega - 180 deg) in degrees)^2.
 * The reference's decoys carry ref2015_cart's per-residue energies (tests/golden/pose_energies.json): on their coordinates this
 * tether is 13 x Rosetta's omega column and the bonded term below 25 x its cart_bonded column.  Round-4 scans on 2 x 1024 decoys
 * per map (profiles/README.md; profiles/history/runs_r01_r04.sh.txt section r04_model_scan*.sh): softer constants HERE (0.02 with 0.4 x the bonded stiffness) improve
 * the default protocol's outcome a little (X-ray decoys within 1 A 83 -> 88 %) but let --no-fastrelax decoys twist (peptides beyond
 * 60 degrees 11 -> 24 % on the X-ray map); applied in the relax stage alone -- whose weights are ref2015_cart's, the score function
 * the comparison is about -- they give most of the gain and leave the centroid stage's calibration alone.  So the constants stay
 * and protocol.SF_FA_SCALE carries the correction. */
#ifndef TRX2_OMEGA_K
#define TRX2_OMEGA_K 0.05
#endif

/* cart_bonded surrogate (Cartesian runs only): E = K (x - x0)^2 on bond lengths, bond angles and two impropers (CB
 * chirality, carbonyl planarity) around the ideal values above, weight 0.1 in sf_cart (folding/data/scorefxn_cart.wts).
 * Stiffness is CALIBRATED so that --no-fastrelax decoys reproduce the geometry spread measured on the reference's decoys
 * (tools/derive_constants.py: bonds sd 0.009-0.011 A, N-CA-C sd 2.4 deg, C-N-CA sd 2.0 deg, omega sd 6.9 deg): with these
 * values tests/diag/scratch_r01_r04/model_scan.py gives bonds 0.007-0.008 A, N-CA-C 2.6-2.8, C-N-CA 2.5-2.9, omega 4.7-7.0.  Softer angles
 * (K 80: spread 8.9 deg) let the chain cheat on the restraints; stiffer ones (K 8000) push the strain back into omega.
 * TRX2_CART_KSCALE: model scans only. */
#ifndef TRX2_CART_KSCALE
#define TRX2_CART_KSCALE 1.0
#endif
#define TRX2_CART_KLEN (15000.0 * TRX2_CART_KSCALE) /* per A^2   */
#define TRX2_CART_KANG (3000.0 * TRX2_CART_KSCALE)  /* per rad^2 */
#define TRX2_CART_KIMP (300.0 * TRX2_CART_KSCALE)   /* per rad^2 */

/* ---- minimiser (own design; Rosetta's lbfgs_armijo_nonmonotone is not in the tree) --------------- */
/* L-BFGS history (stored correction pairs).  No reference pin exists (Rosetta's value is not in the tree).  12 and 8 give
 * the same outcome on 1024-decoy samples against the reference decoys; 8 needs 3.5 % more evaluations but makes every step's
 * two-loop recursion a third shorter (+5 % decoys/s); 6 starts to cost accuracy on the X-ray map (profiles/README.md). */
#ifndef TRX2_LBFGS_M
#define TRX2_LBFGS_M 8
#endif
#define TRX2_LS_PAST 3       /* non-monotone window */
#define TRX2_LS_C1 1.0e-4
#define TRX2_LS_SHRINK 0.5
#define TRX2_LS_MAXTRIAL 20
#ifndef TRX2_MIN_TOL
#define TRX2_MIN_TOL 1.0e-6  /* folding.py:91 fractional tolerance: 2|f0-f1| <= tol (|f0|+|f1|+eps) */
#endif
/* The first step of a minimiser run.  A cold start (no correction pair yet) is a steepest-descent step of unit length, alpha =
 * min(1, 1/|g|); 26 of the default protocol's 35 runs start from the converged point of a run on nearly the same function (RepeatMover,
 * remove_clash, the FastRelax ramps) and needed 7-10 halvings to get from there to an acceptable step -- 10-20 evaluations for 1-2 accepted
 * iterations in every ramp run (profiles/r05_run_profile_*.txt).  A run flagged TRX2_RUN_WARM that starts in the space (torsion /
 * Cartesian) of the decoy's last stored pair scales its first step by that pair's s.y / y.y (what L-BFGS would have used had the run
 * gone on), still capped at unit length: alpha = min(gamma, 1/|g|).  Own design, as the whole minimiser (Rosetta's is not in the tree);
 * which runs carry the flag is the protocol table's business (protocol.py). */
#define TRX2_RUN_PRECHECK 1 /* trx2_run.precheck bit 0: remove_clash's guard            */
#define TRX2_RUN_WARM 2     /* trx2_run.precheck bit 1: warm first step (see above)     */
#define TRX2_CLASH_BREAK 10.0 /* utils_ros.py:701 */
/* Offset per rama residue used ONLY in remove_clash's guard `rama + vdw < 10` (not in the minimised energy).  Rosetta's rama is
 * negative in favoured regions, so a clash-free pose passes the guard and remove_clash stops; this surrogate is >= 0 (about +1
 * per residue on folded decoys), so without an offset the guard can never fire and all five rounds always run.  Round 4: on the
 * reference's decoys the surrogate sits 1.3 per residue above ref2015's rama_prepro column (tests/test_pose_energies.py) -- the
 * offset that makes the two agree in the mean -- and with the default protocol (relax stage on) the outcome does not move with
 * it (1024 decoys per map: NMR 0.751 / 0.750 A, X-ray 0.477 / 0.479 A) while 4.5 % of the evaluations go.  (Rounds 1-3, without
 * the relax stage, kept 0; --no-fastrelax pays for the offset with its tightest NMR bin: 7 -> 3 % within 0.5 A, medians 0.763 ->
 * 0.773 / 0.501 -> 0.512 A, 7 % fewer evaluations.) */
/* Round 5: the fitted rama term carries Rosetta's own level (class constants fitted to rama_prepro), so the guard needs no offset. */
#ifndef TRX2_RAMA_GUARD_OFFSET
#define TRX2_RAMA_GUARD_OFFSET (TRX2_RAMA_FIT_ON ? 0.0 : -1.3)
#endif
#define TRX2_MAX_RUNS 64

/* ---- backbone-visible part of the full-atom refinement (folding/folding.py:200-268; "a11-lite") ----------------------
 * The reference ends every decoy with FastRelax x 2 on ref2015_cart + atom_pair 5 / dihedral 1 / angle 1 (folding.py:202-205)
 * with restraints RE-SELECTED at PCUT 0.15 (round 1, :230-231) and 0.30 (round 2, :236-237), pairs touching a glycine dropped
 * (add_rst(.., nogly=True), utils_ros.py:713-717).  No full-atom model exists here; what a backbone sees of it:
 * the two re-selections and the re-weighted score, run through the ramps of the two FastRelax scripts
 * (folding/data/1relax_round1.txt: torsion x 2 then Cartesian x 1; 2relax_round2.txt: Cartesian x 2; each four
 * `ramp_repack_min <fa_rep scale> <tolerance> <coordinate-constraint weight: unused, no such constraints> <iterations>`). */
#define TRX2_RELAX_PCUT1 0.15
#define TRX2_RELAX_PCUT2 0.30
/* trx2_run.pair_filter: which selection of the map's restraints a run sees */
#define TRX2_FILTER_ALL 0    /* add_rst at the map's PCUT                                       */
#define TRX2_FILTER_ODR 1    /* ... without the pairs flagged in idr (mode 3, first stage)      */
#define TRX2_FILTER_RELAX1 2 /* re-selected at RELAX_PCUT1, no pair touching a glycine          */
#define TRX2_FILTER_RELAX2 3 /* ... at RELAX_PCUT2                                              */

/* one minimiser run of the staged protocol (folding.py:119,164-171; utils_ros.py:699-703) */
typedef struct trx2_run {
  float w[TRX2_NW]; /* score-function weights                                                  */
  int max_iter;     /* MinMover.max_iter                                                       */
  int sep_lo;       /* restraints with sep_lo <= |a-b| < sep_hi are active (add_rst)           */
  int sep_hi;
  int precheck;     /* bit 0 (TRX2_RUN_PRECHECK): before running, if rama+vdw (raw) < CLASH_BREAK jump to skip_to; bit 1: TRX2_RUN_WARM */
  int skip_to;      /* run index to continue with when the precheck fires                      */
  int cartesian;    /* 1: minimise Cartesian coordinates (MinMover.cartesian(True))            */
  int pair_filter;  /* TRX2_FILTER_*: 1: only restraints of pairs NOT flagged in the map's idr mask (add_idr_rst with the complement,
                       mode 3's first stage, folding.py:173-179); 0: all selected restraints; 2 / 3: the relax re-selections */
  float tol;        /* convergence tolerance of this run (MinMover's / ramp_repack_min's); 0: TRX2_MIN_TOL */
} trx2_run;

#endif
