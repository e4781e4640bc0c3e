/* trx2fold.h -- C ABI of libtrx2fold.so, the MI355X-native replacement of the trRosettaX2-Dynamics folding
 * hot path (one PyRosetta process per decoy -> one batched on-device fold per distogram).
 *
 * Drop-in boundary.  The reference has no FFI for this path: it shells out
 *     python ./folding/folding.py -NPZ .. -FASTA .. -OUT .. {options}
 * once per decoy (/root/reference/utils_trX2dy/utils.py:484-505).  The entry points below are what a binding
 * for that path needs; each cites the reference code it replaces.  Plain pointers and sizes only; all
 * pointers are HOST pointers owned by the caller unless the name says _device.
 *
 * Threading: one ctx = one GPU stream + one distogram ("map").  A ctx is not thread-safe; distinct ctxs may
 * be used concurrently (the NMR and X-ray chains of run_inference.py:310-318 run as two ctxs).  No global
 * state.  Every function returns 0 on success, non-zero on error with text in trx2_last_error(ctx).
 */
#ifndef TRX2FOLD_H
#define TRX2FOLD_H

#include <stdint.h>

#include "trx2_model.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct trx2_ctx trx2_ctx;

/* folding/data/params.json + the -pd flag (folding/utils_ros/arguments.py:11) */
typedef struct trx2_params {
  double ebase;     /* EBASE */
  double erep[3];   /* EREP  */
  double drep[3];   /* DREP  */
  double meff;      /* MEFF  */
  double dcut;      /* DCUT  */
  double alpha;     /* ALPHA */
  double dstep;     /* DSTEP */
  double astep_deg; /* ASTEP */
  double pcut;      /* PCUT / -pd : add_rst selection threshold (utils_ros.py:707) */
} trx2_params;

/* per-decoy status codes */
#define TRX2_OK 0
#define TRX2_DIVERGED 1 /* non-finite energy or coordinates */
#define TRX2_MAXEVAL 2  /* evaluation budget exhausted before the protocol finished */

/* 2 since round 5: trx2_run.precheck became a bit field (bit 0 the remove_clash guard, bit 1 TRX2_RUN_WARM: include/trx2_model.h) and
 * TRX2_MAX_RUNS is 64 (48 under version 1's first builds); exports added in round 4 (shared launches, single-decoy waves) kept their names;
 * round 5 also added trx2_set_shared_launch_halves (an addition: no existing entry point changed with it). */
int trx2_abi_version(void);

/* replaces pyrosetta.init + process start-up (folding/folding.py:48): binds a GPU and creates a stream */
int trx2_ctx_create(int device, trx2_ctx** out);
void trx2_ctx_destroy(trx2_ctx* ctx);
/* lanes = 2: trx2_fold_batch folds batches of 32 or more decoys as two halves on two streams (second half from an internal
 * host thread that lives as long as the second lane: started by this call, parked between folds, joined by lanes = 1 or
 * trx2_ctx_destroy), so that one half's step kernel overlaps the other half's pair kernel: +24 % (distances only) / +32 % (all
 * channels) decoys/s at L=150, B=64 on MI355X.  The reference has no counterpart: its decoys are separate OS processes
 * (utils_trX2dy/utils.py:501-503).  Every decoy keeps its identity (seed, decoy0 + index); results equal those of folding the
 * two halves as separate batches.  lanes = 1 (default) restores one stream.  Measured on MI355X, round 3 (one call of 64 decoys,
 * L=150): two lanes 404 against 344 decoys/s with distances only; two CHAINS of 64 on their own contexts (two streams already)
 * fold faster with one lane each (814 decoys/s at the end of the round; 637 against 495 when compared), L=400 with 32 decoys 90 against 78.  GPU_MAX_HW_QUEUES=8 (HIP runtime) changes
 * nothing at two or three concurrent streams and costs 16 % at four (2 chains x 2 lanes): the library neither needs nor sets it;
 * keep a process at two to three folding streams (pipeline.run_single and fold.fold_arrays choose the lanes accordingly). */
int trx2_ctx_set_lanes(trx2_ctx* ctx, int lanes);
/* Slot pool.  trx2_fold_batch folds its B decoys on min(B, slots) decoy slots; every launch pair serves the slots, and a slot
 * whose decoy has finished takes the next decoy of the batch on the device (no host round trip).  A batch ends with its slowest
 * decoy: with one slot per decoy (slots = 0, the default) a quarter to a third of all launches serve a shrinking set of live
 * decoys; with fewer slots than decoys the slots stay busy until the queue is empty.  The reference's counterpart is its
 * process pool (ThreadPoolExecutor over `python folding.py` children, utils_trX2dy/utils.py:501-503), which also starts the next
 * decoy when a worker frees up.  A decoy is identified by (seed, decoy0 + index) and its results do not depend on the slot that
 * folded it, on `slots`, or on the order of completion.  (A fold that starts on 160 slots or more -- 128 for chains of up to 128 residues -- steps
 * with a low-register instantiation of the step kernel, eight waves per CU instead of four: the same operations in the same order,
 * tested bit for bit.)  How many slots: with the kernels of round 3, as many as the job offers up to ~960 per lane
 * (INTEGRATION.md).  trx2_last_fold_slot_efficiency: sum of evaluations over the decoys of the last fold / (launch pairs x slots). */
int trx2_ctx_set_pool(trx2_ctx* ctx, int slots);
/* Tail of a fold.  Once the queue is empty the slots retire one by one, but a launch over several decoy groups (more than 64 slots)
 * keeps its full length while every group still holds a live decoy.  mode 1 (default): whenever the live decoys fit into one group
 * fewer, those of the last group are moved into retired slots on the device and the launches shrink by a group, and below one group
 * the decoys per wave halve (32, 16, ..), each shape with its own row plan of the pair kernel (how many workgroups a row of the
 * restraint lists is cut into).  A decoy's arithmetic then differs from mode 0 by the ORDER in which the records of a residue are
 * added from the moment the shape changes -- rounding, which thousands of minimiser steps amplify: the same kind of difference
 * as folding the decoy in a batch of another size (measured at L=150, 1280 decoys on 2 x 192 slots: tests/test_gpu_configs.py; how many slots: INTEGRATION.md).
 * mode 2: groups are dropped with the row plan kept and the waves are never narrowed: bitwise equal to mode 0, ~9 % slower per
 * call of 64 decoys.  mode 0: off.  No counterpart in the reference (its decoys are separate processes).
 * Environment (A/B timing and tests only): TRX2_NSPLIT=n cuts every row into n slices whatever the shape; TRX2_ROW_TARGET=t sets
 * the plan's target of list entries per slice and partner residue of a wave step (default 18). */
int trx2_ctx_set_tail_compaction(trx2_ctx* ctx, int mode);
/* Shared launches (process-wide; TRX2_ENGINE_STREAMS=1..3 sets the engines per device, default 2).  Default: on as soon as five
 * contexts are alive in the process (second lanes not counted) -- up to four chains fold 3-28 % faster launching for themselves, one
 * hardware queue each; from the fifth on the queues are shared and the engines win (6 / 8 / 12 folds in flight: 6.3 / 5.3 / 4.5 us per
 * fold-evaluation against 9.7 / 7.3 / 7.1; MI355X, L=150, profiles/history/runs_r01_r04.sh.txt section r04_run23.sh).  TRX2_SHARED_LAUNCH=0 / 1 or mode 0 / 1 below force either.  A fold of ONE decoy -- every feedback iteration of run_inference.py:97-139 is one -- leaves the chip idle: a launch pair
 * of ~24 us on a few workgroups, thousands of them in sequence.  With shared launches such a fold does not launch for itself: it hands
 * its argument blocks (its own map's tables, row lists and row plan; its own state and buffers) to an engine thread of the library,
 * whose launch pairs step the single-decoy folds of ALL contexts that are folding at that moment (k_pair1_multi: blockIdx.z = fold,
 * k_step_multi: blockIdx.y = fold), and sleeps until its decoy has reported.  The caller's side does not change -- one host thread
 * per chain calling trx2_fold_batch, as pipeline.run_batch does for the chains of run_inference.py:339-348's targets -- and neither do
 * the results: a fold's arithmetic does not depend on what shares its launches (bit-identical to mode 0: tests).  The reference's
 * counterpart is its process pool over `python folding.py` children (utils_trX2dy/utils.py:501-503).
 * mode 1 on, 0 off, -1 back to the default rule. */
int trx2_set_shared_launches(int mode);
/* Shape of the shared launches.  A step launch of n folds is n workgroups walking a chain of dependent phases; a pair launch fills the
 * chip; one after the other the chip does pair work half of the time.  In half-evaluation form (k_half_multi) the folds of an engine
 * run in two halves half an evaluation apart, and ONE kernel carries the step role of one half beside the pair role of the other --
 * same arithmetic, bit-identical results (tests/test_gpu_shared_launch.py).  Pays when a pair launch fills the chip: measured on MI355X
 * in batch mode at L=150 (profiles/README.md, round 5) 3 / 8 / 16 / 32 targets in flight -4 % / +1.5 % / +15 % / +6 %.
 * mode 1 on, 0 off, -1 (default) the library's rule: TRX2_ENGINE_HALF if set, otherwise from twelve live contexts on.
 * Serves folds with one wave per row (trx2_ctx_set_single_decoy_waves(ctx, 1)), the segment cache, and chains of up to 256 residues. */
int trx2_set_shared_launch_halves(int mode);
/* Shape of the pair kernel for the context's SINGLE-decoy folds.  waves = 4 (default): one workgroup of four waves per row of the
 * restraint lists -- the shortest evaluation while few folds are in flight (run_inference.py on one target: two chains).  waves = 1:
 * one wave per row (k_pair1): a row's ~70 entries and the chain's partner residues in 2-3 steps of one wave, no workgroup barrier,
 * the sums straight out of the wave -- a quarter of the waves per fold, which is what counts when the folds of MANY chains share
 * launches (batch mode).  Measured on MI355X, L=150, all channels, default protocol: one target (two chains) 26.1 against 21.7
 * decoys/s; sixteen targets in flight 97 against 109; 64 folds per launch 405 k against 619 k fold-evaluations/s.  The two shapes add a
 * residue's terms in different orders: results differ by rounding, as between batches of different widths; within one shape a
 * fold's result does not depend on what shares its launches.  pipeline.run_batch sets 1, everything else keeps 4. */
int trx2_ctx_set_single_decoy_waves(trx2_ctx* ctx, int waves);
/* measurement helpers: out[9] = chunks of launch pairs the device's engines enqueued, folds x chunks (ratio: folds per launch), folds
 * completed, seconds their host threads spent enqueuing, seconds they waited for the GPU; and, while trx2_set_shared_launch_profiling(1)
 * is on (one launch pair per chunk of 16 bracketed by HIP events on the engine's stream): summed milliseconds of the sampled pair and
 * step launches, the number of samples, the folds they held (in half-evaluation form: of the sampled evaluation's first and second
 * half launch -- pair role of one half of the folds beside the step role of the other, then the reverse) */
int trx2_shared_launch_stats(int device, double* out);
int trx2_set_shared_launch_profiling(int on);
int trx2_last_fold_slot_efficiency(trx2_ctx* ctx, double* eff);
const char* trx2_last_error(const trx2_ctx* ctx);

/* replaces np.load(NPZ) + gen_rst + add_rst selection (folding/folding.py:56-63; utils_ros.py:6-146,706-723).
 * dist[L][L][37], omega[L][L][25], theta[L][L][25], phi[L][L][13] float32 row-major; omega/theta/phi NULL
 * <=> --no-orient.  seq[L] one-letter codes.  Builds the spline tables on the device (kernel K2); they stay
 * resident for every later fold/eval on this ctx. */
int trx2_set_map(trx2_ctx* ctx, int L, const char* seq, const float* dist, const float* omega, const float* theta,
                 const float* phi, const trx2_params* prm);
/* same with DEVICE pointers (in-memory hand-off from the trX2 front-end, utils_trX2dy/utils.py:783-796) */
int trx2_set_map_device(trx2_ctx* ctx, int L, const char* seq, const float* dist_dev, const float* omega_dev,
                        const float* theta_dev, const float* phi_dev, const trx2_params* prm);

/* The other restraint builders of folding/folding.py:60-68 (-r) and the pair mask of mode 3.
 * trx2_set_map_ex: idr[L][L] (or NULL) flags "disordered" pairs (npz['idr'] of the reference).  rst_kind 0 = gen_rst,
 *   1 = gen_idp_rst (utils_ros.py:196-373: flagged pairs are normalised by their most probable bin).  With an idr mask set, runs
 *   whose trx2_run.pair_filter is 1 use only the restraints of UNflagged pairs (add_idr_rst with 1 - idr, folding.py:173-179).
 * trx2_set_map_af2: gen_rst_af2 (utils_ros.py:148-194): dist64[L][L][64] AlphaFold-style distogram, edges63[63] its bin edges;
 *   C-alpha -- C-alpha restraints with 60 knots; distances only (the reference raises on --orient).
 * trx2_override_table_rows: replaces the VALUES of n rows of one channel's table (y[n][K], as they would be printed; K = 35 or 60 /
 *   28 / 28 / 16) and recomputes their spline; pairs (a[i], b[i]), a < b for dist / omega.  gen_gpcr_rst (utils_ros.py:484-654) =
 *   gen_rst + its ling_sumlt edits of the flagged pairs, which the caller computes from the known structures
 *   (trrosettax2-dynamics_amd/restraints.py). */
int trx2_set_map_ex(trx2_ctx* ctx, int L, const char* seq, const float* dist, const float* omega, const float* theta,
                    const float* phi, const trx2_params* prm, const unsigned char* idr, int rst_kind);
int trx2_set_map_af2(trx2_ctx* ctx, int L, const char* seq, const float* dist64, const double* edges63, const trx2_params* prm);
int trx2_override_table_rows(trx2_ctx* ctx, int channel, int n, const int* a, const int* b, const double* y);

/* read the tables back (parity tests).  channel 0 dist, 1 omega, 2 theta, 3 phi.
 * y_y2: [L][L][K][2] float (value, second derivative), K = 35/28/28/16 (60 for the distance channel of an af2 map); knots: [K];
 * prob: [L][L];
 * gen, sel: [L][L] bit masks (TRX2_M_*).  Any output pointer may be NULL. */
int trx2_get_tables(trx2_ctx* ctx, int channel, float* y_y2, float* knots, float* prob, unsigned char* gen,
                    unsigned char* sel);

/* one energy+gradient evaluation of B decoys through the fold's own kernels (K1 NeRF, K3/K4 pair terms,
 * K5 torsion gradient) -- what one ScoreFunction(pose) + derivative call does (folding/folding.py:74-84).
 * tors[B][L][3] (phi, psi, omega; radians); w[TRX2_NW]; restraints with sep_lo <= |a-b| < sep_hi.
 * Outputs (each may be NULL): e_terms[B][TRX2_NTERMS] raw energies, f_total[B] weighted sum,
 * grad[B][L][3], xyz[B][L][5][3] (N CA C O CB). */
int trx2_eval_batch(trx2_ctx* ctx, int B, const float* tors, const float* w, int sep_lo, int sep_hi,
                    double* e_terms, double* f_total, float* grad, float* xyz);

/* replaces `repeat` runs of folding.py (set_random_dihedral .. remove_clash, folding/folding.py:109-171):
 * folds B decoys of the current map with the staged protocol runs[nruns] (trx2_run, trx2_model.h).
 * Start torsions: tors0[B][L][3] if non-NULL, else the reference's random table drawn from (seed, decoy0+i).
 * Outputs (each may be NULL): tors_out[B][L][3], xyz_out[B][L][5][3], e_terms[B][TRX2_NTERMS] and f_final[B]
 * under the last run's weights, status[B], n_evals[B], n_iters[B]. */
int trx2_fold_batch(trx2_ctx* ctx, int B, const trx2_run* runs, int nruns, uint64_t seed, uint32_t decoy0,
                    const float* tors0, int max_evals, float* tors_out, float* xyz_out, double* e_terms,
                    double* f_final, int* status, int* n_evals, int* n_iters);

/* Feedback step between folds on the device (the caller on both sides of the fold).
 * trx2_feedback_bins replaces get_neighbors + the one-hot binning of utils_trX2dy/utils.py:125-235,294-316: from the
 * backbone of one decoy (xyz[L][5][3] N CA C O CB as read from its PDB file, NaN = absent; seq[L] one-letter, 'G' gets the
 * virtual C-beta) to the realised bin of every ordered pair: jd (distance, 0 = no contact), jo, jt and jp -- jp from THETA on
 * phi's edges, as the reference does (utils.py:226).  The edge arrays are passed in so that they are the very doubles numpy
 * builds: np.arange(2, 20.5, 0.5), np.arange(-pi, pi, pi/12), np.arange(0, pi, pi/12); dmax = 20.  Outputs [L][L] int8.
 * trx2_feedback_process replaces process_distribution_with_pred_distribution (utils.py:379-403, flag "0HD": mask
 * max_k p < 0.5, decay 0.5 above 0.05, renormalise, Gaussian filter along the bins) for one channel in[L][L][K] with the
 * bins of that channel; w9 = the nine weights of scipy's gaussian kernel for sigma (radius 4); norm = 0 gives the
 * un-normalised cumulative `tmp` array.  All pointers are host memory; the context's stream and scratch are used. */
int trx2_feedback_bins(trx2_ctx* ctx, int L, const char* seq, const float* xyz, const double* d_edges, int nd,
                       const double* a_edges, int na, const double* p_edges, int np_, double dmax, signed char* jd,
                       signed char* jo, signed char* jt, signed char* jp);
int trx2_feedback_process(trx2_ctx* ctx, int L, int K, const float* in, const signed char* bins, const double* w9,
                          int norm, int smooth, float* out);

/* The same step on the distograms RESIDENT in the context (those of the last trx2_set_map, or of the previous step): bins of
 * the decoy, the cumulative `tmp` array, the re-weighted channels (dist only, or all four with angle != 0) and new restraint
 * tables -- everything run_inference.py:75-131 does between two folds.  Only the decoy's coordinates go in;
 * max_tmp_change = max |tmp_new - tmp_old|, the reference's convergence measure (stop below 0.01, run_inference.py:133).
 * trx2_get_map downloads a resident array: channel 0..3 = dist, omega, theta, phi; 4 = tmp. */
int trx2_feedback_step(trx2_ctx* ctx, const char* seq, const float* xyz, const double* d_edges, int nd, const double* a_edges,
                       int na, const double* p_edges, int np_, double dmax, const double* w9, int angle, float* max_tmp_change);
int trx2_get_map(trx2_ctx* ctx, int channel, float* out);

/* GloCon matrix for clustering (get_glocon_matrix, utils_trX2dy/utils.py:543-569): for every pair of n
 * decoys of the same length, the sum over the upper triangle of |dist6d_p - dist6d_q| with differences <= 3 A dropped,
 * divided by L (L - 1) / 2; dist6d = C-beta distances within dmax (20), 0 elsewhere, as get_neighbors builds them.
 * seqs = n * L one-letter codes (glycines get the virtual C-beta), xyz[n][L][5][3] as read from the PDB files, out[n][n]. */
int trx2_glocon_matrix(trx2_ctx* ctx, int n, int L, const char* seqs, const float* xyz, double dmax, double* out);

/* Reliability score of n decoys, the quantity the reference ranks its initial decoys by (calculate_reliability_score,
 * utils_trX2dy/utils.py:352-372; run_inference.py:60-73): xyz[n][L][5][3] as read from the decoys' PDB files ->
 * counts[n][2] = (residues with a phi/psi pair as Biopython's PPBuilder reports them, those among them with phi <= 0).  The score
 * is counts[.][1] / counts[.][0] (0 when there is none). */
int trx2_reliability_scores(trx2_ctx* ctx, int n, int L, const float* xyz, int* counts);

/* Batched optimal superposition: C-alpha RMSD and TM-score of every pair between two sets of structures with the
 * same L ALIGNED residues -- what the reference obtains by running its prebuilt `TMscore` binary once per pair
 * (utils_trX2dy/utils.py:514-541 for clustering, evaluate_utils.py:33-100 for evaluation).  xa[n][L][3], xb[m][L][3] C-alpha
 * coordinates as read from the PDB files; xb NULL = xa against itself (symmetric n x n).  l_norm: the TM-score normalisation
 * length (<= 0: L).  rmsd[n][m], tm[n][m]; either may be NULL.  RMSD by Horn's quaternion form of the Kabsch problem; TM-score by
 * the TM-score program's seeded iterative search, one wave per (pair, seed fragment), in float64. */
int trx2_superpose_matrix(trx2_ctx* ctx, int n, int m, int L, const float* xa, const float* xb, double l_norm, double* rmsd, double* tm);

/* measurement helper (bench.py roofline leg): replays the pair-energy kernel n_rep times on the ctx stream
 * for the coordinates of the last eval/fold batch and returns the average launch duration in milliseconds
 * measured with hipEvents on that stream, plus the number of selected term-evaluations per launch. */
int trx2_time_pair_kernel(trx2_ctx* ctx, int B, const float* w, int sep_lo, int sep_hi, int n_rep, double* ms_avg,
                          double* term_evals);

/* measurement helpers (bench.py): with every > 0, every `every`-th evaluation of a fold on this ctx is bracketed by HIP events
 * on the ctx stream (before the pair kernel | between | after the step kernel); trx2_last_fold_kernel_times returns the
 * averages over the sampled evaluations of the last fold (samples beyond four medians of their kernel -- a host stall between an event
 * and the launch behind it -- dropped) -- the live launch durations of both kernels over a whole fold, not only on final coordinates.  Sampling inserts event records into the stream: use it in an untimed fold.
 * trx2_ctx_info: layout facts the roofline arithmetic needs (key TRX2_INFO_*), valid after a fold/eval on the ctx. */
#define TRX2_INFO_GROUP_WIDTH 0 /* decoys per wave of the pair kernel */
#define TRX2_INFO_SLAB_BYTES 1  /* bytes of pair-kernel records the step kernel sums per residue (average over residues) */
#define TRX2_INFO_PAIR_WGS 4    /* workgroups of one pair-kernel launch */
#define TRX2_INFO_CART_STAGED 5 /* stored L-BFGS pairs the Cartesian role stages in LDS at the current chain length (0: none) */
#define TRX2_INFO_LBFGS_M 2     /* stored correction pairs */
#define TRX2_INFO_L 3
int trx2_ctx_set_profiling(trx2_ctx* ctx, int every);
int trx2_last_fold_kernel_times(trx2_ctx* ctx, double* pair_ms_avg, double* step_ms_avg, int* n_samples);
int trx2_ctx_info(const trx2_ctx* ctx, int key, double* value);

/* seconds spent inside the last trx2_fold_batch between first launch and results on host, and the number of
 * pair-kernel launches it made */
int trx2_last_fold_stats(trx2_ctx* ctx, double* seconds, int* n_launches);

#ifdef __cplusplus
}
#endif
#endif
