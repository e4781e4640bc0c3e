// kernel_step.h -- K1/K5/K6: per-decoy step kernels, torsion-space and Cartesian-space roles, and their launchable forms -- included by trx2fold.hip.
// Not a stand-alone header: it relies on the macros, constant tables and helpers defined above its #include.
#pragma once
// =================================================================================================
// K1/K5/K6: per-decoy chain kernel
// =================================================================================================
// PH_REPORT: the decoy's protocol is over; its slot waits for ONE more evaluation, at the accepted point under the last run's
// weights, which is the decoy's report (folding.py prints the final score the same way); then the slot takes the next decoy of
// the queue or retires (PH_DONE).
enum { PH_START = 0, PH_LS = 1, PH_DONE = 2, PH_REPORT = 3 };
enum { MODE_INIT = 0, MODE_STEP = 1, MODE_FINISH = 2 };
// integer state slots
// SI_RUN and SI_SEQ share one aligned 8-byte word: in the fused step launch the two workgroups of a decoy read its state
// while one of them may be writing a run transition; a single 8-byte store / load cannot be seen half-updated, so a
// reader gets (old run, old seq) or (new run, seq of THIS launch -> "already stepped"), never a mixture.
// SI_WSPACE: the space the decoy's last stored correction pair lived in (0 none, 1 torsion, 2 Cartesian): a run that starts in the same
// space takes its first step with that pair's Hessian scale SD_GAMMA instead of 1 / |g| (runs flagged TRX2_RUN_WARM, trx2_model.h)
enum { SI_RUN = 0, SI_SEQ, SI_PHASE, SI_ITER, SI_NLS, SI_HL, SI_HH, SI_NH, SI_STATUS, SI_NEVALS, SI_NITERS, SI_WSPACE, SI_N = 16 };
// double state slots
// SD_GAMMA: s.y / y.y of the newest stored pair = the initial Hessian scaling of the two-loop recursion (torsion role)
enum { SD_F = 0, SD_ALPHA, SD_GD, SD_FH0, SD_FH1, SD_FH2, SD_GAMMA, SD_N = 8 };

struct ChainArgs {
  int L, B, mode, nruns, max_evals;
  const int* seq_ctr;  // evaluation number (device counter bumped by k_pair): a decoy is stepped once per evaluation, by the
                       // torsion OR the Cartesian role (SI_SEQ)
  const trx2_run* runs;
  int* st_i;       // [B][SI_N]
  double* st_d;    // [B][SD_N]
  float* rho;      // [B][LBM]
  double* gram;    // [B][GR_N] scalars of the Gram form of the recursion (chains of one residue per thread)
  float4 *X, *G, *D, *XT;  // [B][L] (phi, psi, omega, -)
  float4 *S, *Y;           // [B][LBM][L]
  float4* P;               // [B][L][5] trial coordinates, decoy-major: N CA C O CB (15 floats + pad) | backbone H, hasH
  float4* xyzT;            // [ngrp][L][5][BW] decoy-minor copy for the pair kernel (its lanes are decoys), atoms in xt_pack order
  int BW;
  float4* geom;            // [B][L][3] internal geometry per residue (ResGeom)
  float* wcur;             // [B][8]
  const float* FA;         // [slice][B][L][24] pair-kernel records: gradient on the six atoms + the pair energies
  const unsigned char* nslice;  // [L] slices the pair kernel cut residue r's row into (the row plan of the launch shape)
  int ns_max;                   // the most slices any row of that plan has: record slices below it exist for every residue (speculative prefetch)
  const unsigned char* hasH;  // [L] residue donates a backbone hydrogen bond (has a predecessor, not proline)
  double* e_last;          // [B][NTERMS] raw terms of the last evaluation
  double* f_last;          // [B]
  float* grad_out;         // [B][L][3] (MODE_FINISH)
  int* done_count;         // slots retired
  // ---- the slot pool: B slots fold n_total decoys; a slot whose decoy has reported takes the next one of the queue
  int* slot_id;            // [B] decoy (0 .. n_total-1) in the slot
  int* next_id;            // queue head
  int n_total;
  unsigned long long seed; unsigned decoy0;   // identity of decoy i: (seed, decoy0 + i), whatever slot folds it
  const float* tors0_all;  // [n_total][L][3] start torsions, or NULL: the reference's random start table
  float4* out_xyz;         // [n_total][L][4] final coordinates (N CA C O CB)
  float4* out_X;           // [n_total][L] final torsions
  double* out_e;           // [n_total][NTERMS] raw terms of the report
  double* out_f;           // [n_total]
  int* out_stat;           // [n_total][4] status, evaluations, accepted iterations, slot
};

// Sum of the pair kernel's records of residue r of decoy dec over the nsplit slices of its row (24 floats: gradient on N CA C O
// CB H, then the energies dist omega theta phi vdw hb).  Fixed order, slice by slice: deterministic.
// The first NPRE slices came with the caller's early requests (pre[k] = slice k; a slice the row does not have was read from a valid
// address and is not added).  Round 5 measured what the slice count's round trip costs -- nslice[r] arrives with the decoy's state, the
// records of slices >= 1 are requested only then -- by requesting FOUR slices up front (72 more registers): step kernel 18.99 -> 19.47 us
// for one decoy, 22.78 -> 22.74 us at 32 per launch, 33.43 -> 33.29 us at L = 400 (tools/step_ab.py, profiles/r05_step_ab.txt): nothing.
// The step is bound by the ~8000 vector instructions each of its waves issues alone on its SIMD, not by its round trips; the phase
// stamps (profiles/r05_stamp_step_*.txt) charge a phase the latency it WAITS for, which the unstamped kernel overlaps.  One slice, as before.
#define FA_NPRE 1
__device__ __forceinline__ void add_record(const float4 (&v)[6], float (&g)[PR_NCOMP], float (&e)[6]) {
  g[0] += v[0].x; g[1] += v[0].y; g[2] += v[0].z; g[3] += v[0].w; g[4] += v[1].x; g[5] += v[1].y; g[6] += v[1].z; g[7] += v[1].w;
  g[8] += v[2].x; g[9] += v[2].y; g[10] += v[2].z; g[11] += v[2].w; g[12] += v[3].x; g[13] += v[3].y; g[14] += v[3].z; g[15] += v[3].w;
  g[16] += v[4].x; g[17] += v[4].y; e[0] += v[4].z; e[1] += v[4].w; e[2] += v[5].x; e[3] += v[5].y; e[4] += v[5].z; e[5] += v[5].w;
}
__device__ __forceinline__ void load_record(const float* FA, int sl, int B, int L, int dec, int r, float4 (&v)[6]) {
  const float4* f = reinterpret_cast<const float4*>(FA + (((size_t)sl * B + dec) * L + r) * PR_REC);
#pragma unroll
  for (int q = 0; q < 6; q++) v[q] = f[q];
}
template <int NPRE>
__device__ __forceinline__ void sum_pair_records(const float* FA, int nsplit, int B, int L, int dec, int r, float (&g)[PR_NCOMP], float (&e)[6], const float4 (&pre)[NPRE > 0 ? NPRE : 1][6]) {
#pragma unroll
  for (int i = 0; i < PR_NCOMP; i++) g[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < 6; i++) e[i] = 0.0f;
#pragma unroll
  for (int k = 0; k < NPRE; k++)
    if (k < nsplit) add_record(pre[k], g, e);
  // the rest two slices per trip, both requested before either is added; the additions keep their order, slice by slice
  for (int sl = NPRE; sl < nsplit; sl += 2) {
    const bool two = sl + 1 < nsplit;
    float4 v[6], u[6];
    load_record(FA, sl, B, L, dec, r, v);
    load_record(FA, two ? sl + 1 : sl, B, L, dec, r, u);
    add_record(v, g, e);
    if (two) add_record(u, g, e);
  }
}
// backbone H from C of the previous residue, N, CA: in-plane bisector (trx2_model.h; oracle: orc_place_h)
__device__ __forceinline__ f3 place_h(f3 Cp, f3 N, f3 CA) {
  const f3 u = unit(N - Cp) + unit(N - CA);
  return N + u * ((float)TRX2_HB_B_NH * rsqrtf(dot(u, u)));
}

// ---- rama / omega terms (trx2_model.h: six-basin prior + the surfaces fitted to the reference decoys' energy tables; oracle: rama_term,
// omega_term).  The per-residue parameter block of the rama term (TRX2_RAMA_NPAR floats = 3 float4) lives behind the map's hasH bytes.
__device__ __forceinline__ const float4* rama_par_ptr(const unsigned char* hasH, int L) {
  return reinterpret_cast<const float4*>(hasH + (((size_t)L + 255) & ~(size_t)255));
}
// energy of one residue from sin / cos of phi, psi; gph, gps = dE/dphi, dE/dpsi (unweighted)
__device__ __forceinline__ float rama_eval(float sph, float cph, float sps, float cps, const float4 (&rp)[3], float& gph, float& gps) {
  float s = 0, dph = 0, dps = 0, h = 0, dhph = 0, dhps = 0;
#pragma unroll
  for (int j = 0; j < TRX2_RAMA_NB; j++) {
    // sin / cos of (phi - phi_k), (psi - psi_k) by the angle-addition identities: 2 sincosf per residue, not 12
    const float sk = c_rama_sc[j * 4], ck = c_rama_sc[j * 4 + 1], tk = c_rama_sc[j * 4 + 2], uk = c_rama_sc[j * 4 + 3];
    const float sa = sph * ck - cph * sk, ca = cph * ck + sph * sk;
    const float sb = sps * uk - cps * tk, cb = cps * uk + sps * tk;
    const float t = c_rama[j * 3 + 2] * expf((float)TRX2_RAMA_KAPPA * (ca + cb - 2.0f));
    const float ta = t * (float)TRX2_RAMA_KAPPA * sa, tb = t * (float)TRX2_RAMA_KAPPA * sb;
    s += t; dph -= ta; dps -= tb;
    if (TRX2_RAMA_FIT_ON && (j == 3 || j == 4)) { h += t; dhph -= ta; dhps -= tb; }   // the two right-handed helical basins
  }
  const float inv = 1.0f / (s + (float)TRX2_RAMA_FLOOR);
  float E = -logf((s + (float)TRX2_RAMA_FLOOR) * (1.0f / (float)TRX2_RAMA_PREF));
  gph = -dph * inv; gps = -dps * inv;
  if (TRX2_RAMA_FIT_ON) {
    const float cm = cph * cps + sph * sps, sm = sph * cps - cph * sps, cp = cph * cps - sph * sps, sp = sph * cps + cph * sps;
    const float r = h * inv;
    const float p0 = rp[0].x, p1 = rp[0].y, p2 = rp[0].z, p3 = rp[0].w, p4 = rp[1].x, p5 = rp[1].y, p6 = rp[1].z, p7 = rp[1].w, p8 = rp[2].x, p9 = rp[2].y;
    E += p0 + p1 * cps + p2 * sps + p3 * cm + p4 * sm + p5 * cph + p6 * sph + p7 * cp + p8 * sp + p9 * r;
    gph += -p3 * sm + p4 * cm - p5 * sph + p6 * cph - p7 * sp + p8 * cp + p9 * (dhph - r * dph) * inv;
    gps += -p1 * sps + p2 * cps + p3 * sm - p4 * cm - p7 * sp + p8 * cp + p9 * (dhps - r * dps) * inv;
  }
  return E;
}
// energy of one peptide's omega tether from sin / cos of psi_i and omega_i (radians); gps, gom = dE/dpsi, dE/domega (unweighted)
__device__ __forceinline__ float omega_eval(float sps, float cps, float om, float& gps, float& gom) {
  if (!TRX2_OMEGA_FIT_ON) {
    float dw = om - TRX2_PI_F;
    dw -= 2.0f * TRX2_PI_F * rintf(dw * (0.5f / TRX2_PI_F));
    dw *= (1.0f / TRX2_DEG_F);
    gps = 0.0f; gom = 2.0f * (float)TRX2_OMEGA_K * dw * (1.0f / TRX2_DEG_F);
    return (float)TRX2_OMEGA_K * dw * dw;
  }
  constexpr float q[9] = TRX2_OMEGA_FIT;
  float x = om - TRX2_PI_F;
  x -= 2.0f * TRX2_PI_F * rintf(x * (0.5f / TRX2_PI_F));
  x *= (0.1f / TRX2_DEG_F);
  const float A = q[0] + q[1] * cps + q[2] * sps, B = q[3] + q[4] * cps + q[5] * sps, C = q[6] + q[7] * cps + q[8] * sps;
  constexpr float st = (float)TRX2_OMEGA_STIFF;
  gom = st * (B + 2.0f * C * x) * (0.1f / TRX2_DEG_F);
  gps = (-q[1] * sps + q[2] * cps) + st * ((-q[4] * sps + q[5] * cps) + (-q[7] * sps + q[8] * cps) * x) * x;
  return A + st * (B + C * x) * x;
}
// peptides that carry the tether: all of them, as in rounds 1-4.  (Rosetta scores none on a terminus -- the reference decoys' tables show ~0
// for the first peptide whatever its angle, conf_1_1's is cis -- and the fit leaves that sample out; but with the first peptide free half of
// this model's decoys twist it beyond 60 degrees, where seven of the reference's eight keep it trans: profiles/r05_model_scan2.txt.)
__device__ __forceinline__ bool omega_on(int r, int L) { return r < L - 1; }

#ifdef TRX2_DBG
__device__ double g_dbg[256][12];  // diagnostic build only: decoy 0's line-search record per evaluation (tools/dbg_linesearch.py)
#endif
#ifdef TRX2_SELFCHECK
__device__ unsigned long long g_selfcheck[6];  // torsion role: checks, mismatches; Cartesian role: checks, mismatches; run starts, starts whose fh[0] != f
#endif
// workgroup barrier of an NW-wave role.  One wave: its LDS operations execute in program order, so only the compiler has
// to be kept from reordering them.
template <int NW>
__device__ __forceinline__ void bsync() {
  if (NW > 1) __syncthreads();
  else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
}
// Sum over the workgroup, total in every thread.  Consecutive calls alternate between two LDS buffers (`flip`), so one
// barrier per call is enough: a wave can be at most one call ahead of the slowest, and then it writes the OTHER buffer.
// (With one buffer every call needed a second barrier just to protect the previous call's reads; the two-loop recursion
// makes 2 x LBM dependent calls per step.)  Every wave must make the same sequence of calls.
#define SBUF_K 12 /* values per wave in the reduction buffers (the energy reduction carries 9) */
template <int K, int NW>
__device__ __forceinline__ void block_sum_n(double (&v)[K], double* s_buf /* [2][NW*SBUF_K] */, int& flip) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; k++) v[k] = wave_sum(v[k]);
  if (NW == 1) return;  // the in-wave sum leaves the total in every lane
  double* buf = s_buf + flip * (NW * SBUF_K);
  flip ^= 1;
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < K; k++) buf[wave * K + k] = v[k];
  bsync<NW>();
#pragma unroll
  for (int k = 0; k < K; k++) {
    double a = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) a += buf[w * K + k];  // fixed order: deterministic
    v[k] = a;
  }
}

__device__ __forceinline__ float dot3(float4 a, float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }

// Internal geometry of one residue, 3 float4 (what torsion-space moves keep fixed; ideal values: trx2_model.h; after a
// Cartesian run the relaxed values extracted from the coordinates -- the oracle's ORC_NGEOM record):
//   g0 = (|N-CA|, |CA-C|, |C-N'|, angle N-CA-C)   g1 = (angle CA-C-N', angle C-N'-CA', |C-O|, angle CA-C-O)
//   g2 = (dihedral N-CA-C-O minus psi, CB coefficients on (b x c), b, c with b = CA-N, c = C-CA)
struct ResGeom {
  float4 g0, g1, g2;
};
__device__ __forceinline__ ResGeom ideal_geom() {
  ResGeom g;
  g.g0 = make_float4((float)TRX2_B_N_CA, (float)TRX2_B_CA_C, (float)TRX2_B_C_N, (float)TRX2_A_N_CA_C * TRX2_DEG_F);
  g.g1 = make_float4((float)TRX2_A_CA_C_N * TRX2_DEG_F, (float)TRX2_A_C_N_CA * TRX2_DEG_F, (float)TRX2_B_C_O, (float)TRX2_A_CA_C_O * TRX2_DEG_F);
  g.g2 = make_float4(TRX2_PI_F, (float)TRX2_CB_KA, (float)TRX2_CB_KB, (float)TRX2_CB_KC);
  return g;
}
// local frame of a residue: CA at origin, C on +x, N in the xy plane (y>0 side)
__device__ __forceinline__ void local_atoms(const ResGeom& g, f3& N, f3& CA, f3& C, f3& CB) {
  float sa, ca;
  fast_sincosf(g.g0.w, &sa, &ca);
  CA = mk3(0, 0, 0);
  C = mk3(g.g0.y, 0, 0);
  N = mk3(g.g0.x * ca, g.g0.x * sa, 0);
  f3 b = CA - N, c = C - CA, a = cross(b, c);
  CB = CA + a * g.g2.y + b * g.g2.z + c * g.g2.w;
}

// Diagnostic build only (-DTRX2_STAMP): thread 0 of decoy 0's torsion-role workgroup accumulates s_memtime cycles per phase
// of every STEP launch into g_cstamp (slot 30 = launches, 31 = launches that computed a new direction).  Every stamp drains
// the memory counters first, so a phase is charged the latency of the loads it issued.
#ifdef TRX2_STAMP
__device__ unsigned long long g_cstamp[32];
#define CSTAMP_DECL unsigned long long cst_prev = 0; const bool cst_on = (dec == 0 && A.mode == MODE_STEP && threadIdx.x == 0); \
  if (cst_on) { __builtin_amdgcn_s_waitcnt(0); cst_prev = __builtin_amdgcn_s_memtime(); }
#define CSTAMP(k) if (cst_on) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
  __builtin_amdgcn_s_waitcnt(0); atomicAdd(&g_cstamp[k], t_ - cst_prev); __builtin_amdgcn_s_waitcnt(0); cst_prev = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define CCOUNT(k) if (cst_on) atomicAdd(&g_cstamp[k], 1ull);
#define KSTAMP_DECL unsigned long long cst_prev = 0; const bool cst_on = (dec == 0 && threadIdx.x == 0); \
  if (cst_on) { __builtin_amdgcn_s_waitcnt(0); cst_prev = __builtin_amdgcn_s_memtime(); }
#else
#define KSTAMP_DECL
#define CSTAMP_DECL
#define CSTAMP(k)
#define CCOUNT(k)
#endif
// L-BFGS history of the decoy staged in LDS for the duration of one step: [LBM][s | y][L] float4, dynamic shared memory
// (HIST_LDS_BYTES, only the one-residue-per-thread instantiations; gfx950 has 160 KB of LDS per CU).  The two-loop
// recursion is 2 x LBM DEPENDENT rounds; read from global memory every round exposed an L2 round trip (~500-700 of its
// ~1000 cycles; the compiler turns a register prefetch into a wait on the load just issued).  Instead the whole history is
// requested at the top of the step with LDS-DMA loads (global_load_lds_dwordx4: no registers, nothing waits on them until
// the recursion starts a phase later) and every round reads the thread's own slot from LDS.
extern __shared__ float4 s_hist[];
#define HIST_LDS_BYTES(L) (LBM * 2 * (L) * 16)  /* the torsion role's staged history: [LBM][s | y][L] float4 */
#define CART_HIST_BYTES(L) ((size_t)(L) * 128)  // one stored pair of the Cartesian role in LDS: [s | y][4][L] float4
#define CART_ARRAYS_BYTES(L) ((((size_t)(L) * 100) + 15) / 16 * 16)  // the Cartesian role's own LDS arrays, in front of its staged pairs
__device__ __forceinline__ void lds_dma16(const float4* src /* per lane */, float4* dst_wave /* wave-uniform: lane i lands at dst + i */) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)dst_wave, 16, 0, 0);
}
// ---- The two-loop recursion from Gram matrices ("vector-free" L-BFGS, Chen et al. 2014), chains of one residue per thread.
// The recursion is 2 x LBM DEPENDENT rounds of { dot with a stored vector, workgroup sum, axpy }: ~400 ns each, 6.5 us of an
// 18 us step (launch-length histogram of single-decoy folds, profiles/README.md) however cheap the sum itself is made.  But
// every dot product in it is a combination of the scalars s_i.y_j, y_i.y_j, s_i.g and y_i.g:
//   s_k . q      = s_k.g - sum_{m newer than k} alpha_m s_k.y_m                         (first loop)
//   y_k . r      = gamma (y_k.g - sum_m alpha_m y_k.y_m) + sum_{m older than k} c_m s_m.y_k      (second loop, c = alpha - beta)
//   direction    = -(gamma g - gamma sum_m alpha_m y_m + sum_m c_m s_m),  g.d from the same scalars.
// So the decoy keeps those scalars (GR_N doubles, pairs in AGE order, 0 = newest); an accepted step costs ONE fused
// workgroup reduction of the GV_N products that involve the new pair and the new gradient, a shift of the matrices, the
// recurrences in registers (every thread, ~250 f64 operations, no barrier) and one pass over the stored vectors.  s_i.g and
// y_i.g follow the gradient by g_new = g_old + y_new (a stored pair lives LBM steps: at most LBM - 1 such updates).
#define GR_SY 0                         /* [LBM][LBM] s_i . y_j */
#define GR_YY (LBM * LBM)               /* [LBM][LBM] y_i . y_j */
#define GR_SG (2 * LBM * LBM)           /* [LBM] s_i . g        */
#define GR_YG (2 * LBM * LBM + LBM)     /* [LBM] y_i . g        */
#define GR_N (2 * LBM * LBM + 2 * LBM)
// products of one accepted step (s, y = the new pair, g = the new gradient, k = the stored pairs by age BEFORE the step)
enum { GV_A = 0 /* s_k.y [LBM] */, GV_B = LBM /* y_k.y [LBM] */, GV_C = 2 * LBM /* s.y_k [LBM-1] */, GV_SY = 3 * LBM - 1, GV_SS, GV_YY, GV_SG /* s.g */,
       GV_YG /* y.g */, GV_GG /* g.g */, GV_N };
static_assert(LBM == 8, "the Gram recursion below is written for 8 stored pairs (thread <-> matrix entry maps, 128 + 16 threads)");
template <int NT>
struct GramLds {
  double gram[GR_N];
  double out[GV_N + 1];
  float part[GV_N * (NT / 16)];
};
// Workgroup sums of GV_N per-thread products: f32 butterfly inside each row of 16 lanes (one DPP-fused add per stage), the
// NT / 16 row sums of a value added in f64 in a fixed order by one thread.  (One f64 DPP sum per value, as block_sum_n does
// it, is ~150 cycles each: 4000 cycles for this set.)
template <int NT>
__device__ __forceinline__ void gram_reduce(float (&pv)[GV_N], GramLds<NT>& gl) {
  constexpr int ROWS = NT / 16;
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < GV_N; k++) {
    float v = pv[k];
    v += dpp_move<0xB1>(v); v += dpp_move<0x4E>(v); v += dpp_move<0x141>(v); v += dpp_move<0x140>(v);
    pv[k] = v;
  }
  if ((tid & 15) == 0)
#pragma unroll
    for (int k = 0; k < GV_N; k++) gl.part[k * ROWS + (tid >> 4)] = pv[k];
  __syncthreads();
  if (tid < GV_N) {
    double a = 0;
#pragma unroll
    for (int r = 0; r < ROWS; r++) a += (double)gl.part[tid * ROWS + r];
    gl.out[tid] = a;
  }
  __syncthreads();
}
// The same with the step's ENERGY riding along (round 5): the weighted total of a trial point is needed before the Armijo test and the
// GV_N products after it -- two workgroup reductions back to back on the critical path of every accepted step.  The products do not depend
// on the test's outcome, so a line-search evaluation computes them up front and reduces both at once: the f64 energy by DPP inside each wave
// and through the alternating buffers of block_sum_n (same operations, same order: the same bits as block_sum_n<1>), the products as above,
// ONE barrier in front of both.  A rejected trial has computed its products in vain (5-10 % of the steps).
template <int NT>
__device__ __forceinline__ double gram_reduce_with_energy(float (&pv)[GV_N], double ft, GramLds<NT>& gl, double* s_buf, int& flip) {
  constexpr int ROWS = NT / 16, NW = NT / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  ft = wave_sum(ft);
#pragma unroll
  for (int k = 0; k < GV_N; k++) {
    float v = pv[k];
    v += dpp_move<0xB1>(v); v += dpp_move<0x4E>(v); v += dpp_move<0x141>(v); v += dpp_move<0x140>(v);
    pv[k] = v;
  }
  double* buf = s_buf + flip * (NW * SBUF_K);
  flip ^= 1;
  if (NW > 1 && lane == 0) buf[wave] = ft;
  if ((tid & 15) == 0)
#pragma unroll
    for (int k = 0; k < GV_N; k++) gl.part[k * ROWS + (tid >> 4)] = pv[k];
  __syncthreads();
  if (NW > 1) {
    double a = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) a += buf[w];  // fixed order: deterministic
    ft = a;
  }
  if (tid < GV_N) {
    double a = 0;
#pragma unroll
    for (int r = 0; r < ROWS; r++) a += (double)gl.part[tid * ROWS + r];
    gl.out[tid] = a;
  }
  __syncthreads();
  return ft;
}
// After gram_reduce: the decoy's scalars follow the step.  `stored`: the pair (s, y) becomes pair 0, every older pair moves
// one place (the oldest drops out); otherwise only the gradient changed.  old0 = this thread's matrix entry before the step
// (entry tid of gram[], tid < 2 LBM^2), old1 = entry 2 LBM^2 + tid (tid < 2 LBM).  Ends with a barrier.
template <int NT>
__device__ __forceinline__ void gram_advance(GramLds<NT>& gl, bool stored, double old0, double old1) {
  const int tid = threadIdx.x;
  const double* o = gl.out;
  if (stored) {
    if (tid < 2 * LBM * LBM) {
      const int i = (tid >> 3) & 7, j = tid & 7;
      if (i < LBM - 1 && j < LBM - 1) gl.gram[tid + LBM + 1] = old0;  // [i][j] -> [i+1][j+1]
    }
    if (tid < LBM) gl.gram[GR_SY + tid] = tid == 0 ? o[GV_SY] : o[GV_C + tid - 1];                       // s_0 . y_j
    else if (tid < 2 * LBM) { if (tid > LBM) gl.gram[GR_SY + (tid - LBM) * LBM] = o[GV_A + tid - LBM - 1]; }  // s_i . y_0
    else if (tid < 3 * LBM) gl.gram[GR_YY + tid - 2 * LBM] = tid == 2 * LBM ? o[GV_YY] : o[GV_B + tid - 2 * LBM - 1];
    else if (tid < 4 * LBM) { if (tid > 3 * LBM) gl.gram[GR_YY + (tid - 3 * LBM) * LBM] = o[GV_B + tid - 3 * LBM - 1]; }
    if (tid < LBM - 1) gl.gram[GR_SG + tid + 1] = old1 + o[GV_A + tid];
    else if (tid >= LBM && tid < 2 * LBM - 1) gl.gram[GR_YG + tid - LBM + 1] = old1 + o[GV_B + tid - LBM];
    if (tid == 2 * LBM - 1) { gl.gram[GR_SG] = o[GV_SG]; gl.gram[GR_YG] = o[GV_YG]; }
  } else {
    if (tid < LBM) gl.gram[GR_SG + tid] = old1 + o[GV_A + tid];
    else if (tid < 2 * LBM) gl.gram[GR_YG + tid - LBM] = old1 + o[GV_B + tid - LBM];
  }
  __syncthreads();
}
// The recurrences: n stored pairs (age order), gamma, g.g -> alpha_m (times gamma, sign folded: coefficient of y_m) and c_m
// (coefficient of s_m) of r = gamma g + sum cy_m y_m + sum cs_m s_m, and g.r.  Entries of pairs >= n are never multiplied
// by anything but zero (and are kept finite: zeroed whenever the history restarts).
__device__ __forceinline__ void gram_recursion(const double* G, int n, double gam, double gg, float (&cy)[LBM], float (&cs)[LBM], double& g_r) {
  double al[LBM], c[LBM], t[LBM], rho[LBM];
#pragma unroll
  for (int k = 0; k < LBM; k++) {
    const double d = G[GR_SY + k * (LBM + 1)];
    double r = __builtin_amdgcn_rcp(d);  // 1 / s_k.y_k: hardware estimate + two Newton steps (a division is ~30 instructions)
    r = fma(fma(-d, r, 1.0), r, r); r = fma(fma(-d, r, 1.0), r, r);
    rho[k] = k < n ? r : 0.0;
    double a = G[GR_SG + k];
#pragma unroll
    for (int m = 0; m < k; m++) a -= al[m] * G[GR_SY + k * LBM + m];
    al[k] = k < n ? rho[k] * a : 0.0;
  }
#pragma unroll
  for (int k = 0; k < LBM; k++) t[k] = G[GR_YG + k];
#pragma unroll
  for (int k = 0; k < LBM; k++) {  // y_k.y_m is symmetric: one read serves (k, m) and (m, k)
    t[k] -= al[k] * G[GR_YY + k * (LBM + 1)];
#pragma unroll
    for (int m = k + 1; m < LBM; m++) { const double v = G[GR_YY + k * LBM + m]; t[k] -= al[m] * v; t[m] -= al[k] * v; }
  }
#pragma unroll
  for (int k = LBM - 1; k >= 0; k--) {
    double yr = gam * t[k];
#pragma unroll
    for (int m = k + 1; m < LBM; m++) yr += c[m] * G[GR_SY + m * LBM + k];
    c[k] = k < n ? al[k] - rho[k] * yr : 0.0;
  }
  double a = gam * gg;
#pragma unroll
  for (int m = 0; m < LBM; m++) {
    a += c[m] * G[GR_SG + m] - gam * al[m] * G[GR_YG + m];
    cy[m] = (float)(-gam * al[m]); cs[m] = (float)c[m];
  }
  g_r = a;
}
// random start torsions: set_random_dihedral (utils_ros.py:656-696) with explicit (seed, decoy, residue) hashing
__device__ __forceinline__ uint64_t splitmix64_dev(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ float4 start_torsions(int L, uint64_t seed, uint32_t decoy, int r, const float* tors0 /* [L][3] of this decoy or NULL */) {
  if (tors0) return make_float4(tors0[(size_t)r * 3], tors0[(size_t)r * 3 + 1], tors0[(size_t)r * 3 + 2], 0);
  if (r >= L - 1) return make_float4(180.0f * TRX2_DEG_F, 180.0f * TRX2_DEG_F, TRX2_PI_F, 0);  // the last residue keeps the extended pose (:658)
  const double cum[6] = TRX2_RAND_CUM_INIT;
  const uint64_t hsh = splitmix64_dev(seed ^ splitmix64_dev(((uint64_t)decoy << 32) | (uint32_t)r));
  const double u = (double)(hsh >> 11) * (1.0 / 9007199254740992.0);
  int k = 0;
  while (!(u <= cum[k])) k++;
  return make_float4(c_rama[k * 3], c_rama[k * 3 + 1], TRX2_PI_F, 0);
}

// NT threads step one decoy, RPT residues per thread (RPT * NT >= L).  NT = 256 is what runs.  One wave (NT = 64, RPT = 3 at
// L = 150) makes every reduction and scan barrier-free but was SLOWER on MI355X (73.7 vs 61.2 us per evaluation,
// profiles/README.md): the step is bound by the per-thread chain of dependent arithmetic and loads, which RPT multiplies,
// not by its ~25 barriers.
// LDS both roles of a step kernel use, declared once per kernel: the protocol table and the Gram scalars (GN = the workgroup
// width of the one-residue-per-thread instantiations, 16 = a stub where the recursion keeps its two-loop form)
#define STEP_RUNS_INTS (TRX2_MAX_RUNS * (int)(sizeof(trx2_run) / 4))
template <int RPT, int NT>
__device__ __forceinline__ void chain_body(const ChainArgs& A, const int dec, int* s_runs, GramLds<(RPT == 1) ? NT : 16>& s_gl) {
  constexpr int NW = NT / 64;
  const int L = A.L, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ double s_buf[2 * NW * SBUF_K];
  int flip = 0;
  __shared__ float s_scan[NW * 12];
  __shared__ float s_alpha[LBM];
  __shared__ int s_i[SI_N];
  __shared__ double s_d[SD_N];
  __shared__ float s_rho[LBM];
  __shared__ float s_phi[RPT * NT + 1];

  int* gi = A.st_i + (size_t)dec * SI_N;
  double* gd_ = A.st_d + (size_t)dec * SD_N;
  CSTAMP_DECL
  // Everything the role test needs is requested at once: the decoy's state, the evaluation counter and the protocol table
  // (into LDS: which run applies is known only when the state has arrived, and reading the run from global memory then was a
  // second dependent round trip, the counter a third).
  const int seq = (A.mode == MODE_STEP) ? *A.seq_ctr : -1;
  for (int i = tid; i < A.nruns * (int)(sizeof(trx2_run) / 4); i += NT) s_runs[i] = reinterpret_cast<const int*>(A.runs)[i];
  if (tid < SI_N && tid >= 2) s_i[tid] = gi[tid];
  if (tid == 0) {  // (run, seq) in ONE 8-byte load
    const unsigned long long rs = *reinterpret_cast<const volatile unsigned long long*>(gi);
    s_i[SI_RUN] = (int)(unsigned)(rs & 0xffffffffull); s_i[SI_SEQ] = (int)(unsigned)(rs >> 32);
  }
  if (tid < SD_N) s_d[tid] = gd_[tid];
  if (tid < LBM) s_rho[tid] = A.rho[(size_t)dec * LBM + tid];
  // per-residue constants of the map and of the launch shape, requested with the state: the slice count of this residue's row
  // (the record sums depend on it) and whether the next residue carries a backbone H (the last store of the kernel does)
  int nsl_pre = 0;
  bool hH_next = false;
  // ... and everything of the evaluation this residue consumes that does not depend on the decoy's state: trial torsions and
  // coordinates, the first slice of the pair records, the accepted point with its gradient and direction.  Requested before the
  // role test instead of after it, their round trip runs beside the state's instead of behind it (~2000 cycles of a step).  A
  // workgroup whose role is not the decoy's current one has asked in vain: 13 % of the torsion role's launches.
  float4 e_xt, e_c[6], e_fa[FA_NPRE][6], e_x, e_g, e_dv, e_rp[3];
  e_rp[0] = e_rp[1] = e_rp[2] = make_float4(0, 0, 0, 0);
  e_xt = e_x = e_g = e_dv = make_float4(0, 0, 0, 0);
#pragma unroll
  for (int q = 0; q < 6; q++) {
    e_c[q] = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < FA_NPRE; k++) e_fa[k][q] = make_float4(0, 0, 0, 0);
  }
  if (RPT == 1) {
    const int rc = min(tid, L - 1);
    hH_next = A.hasH[min(tid + 1, L - 1)] != 0;
    if (A.mode != MODE_INIT) {
      nsl_pre = A.nslice[rc];
      const size_t vr = (size_t)dec * L + rc;
      e_xt = A.XT[vr];
      if (TRX2_RAMA_FIT_ON) { const float4* rpp = rama_par_ptr(A.hasH, L) + (size_t)rc * 3; e_rp[0] = rpp[0]; e_rp[1] = rpp[1]; e_rp[2] = rpp[2]; }
      const float4* xp = A.P + vr * 5;
#pragma unroll
      for (int q = 0; q < 5; q++) e_c[q] = xp[q];
      e_c[5] = xp[rc + 1 < L ? 5 : 0];
#pragma unroll
      for (int k = 0; k < FA_NPRE; k++) load_record(A.FA, min(k, A.ns_max - 1), A.B, L, dec, rc, e_fa[k]);   // slices 0 .. FA_NPRE-1 (clamped into the plan's range)
      if (A.mode == MODE_STEP) { e_x = A.X[vr]; e_g = A.G[vr]; e_dv = A.D[vr]; }
    }
  }
  // chains of up to 512 residues (one residue per thread): the history staged in LDS for the step.  (256 < L <= 512 since round 4: 256 L
  // bytes of dynamic LDS, 128 KB at 512 residues, beside 16 KB of static LDS now that the Cartesian role's arrays of that
  // instantiation live in the dynamic buffer too; rounds 2-3 read it from global memory twice per step.)
  constexpr bool HIST_LDS = (RPT == 1 && NT <= 512);
  // One residue per thread (chains of up to 512 residues on up to 512 threads): the recursion in its Gram form.  The 144
  // scalars do not depend on the chain length.  Beyond 256 residues the stored vectors are not staged in LDS (128 KB at 512
  // residues, beside the Cartesian role's static LDS) but read from global memory: the Gram form reads each of them twice
  // per step, the loads of a pass in flight together, where the two-loop form made 2 x 8 DEPENDENT rounds over them
  // (30-40 % of the step at L = 400, profiles/r02_stamp_step_c4.txt).  (Two residues per thread on 256 threads, round 2's
  // shape for 256 < L <= 512, needs ~370 registers in this form -- the fused kernel has 256: 114 spilled -- and carries two
  // residues through every dependent phase; 512 threads with one residue each fit and halve those chains.)
  constexpr bool GRAM = (RPT == 1);
  double gr_old0 = 0, gr_old1 = 0;  // this thread's entries of the decoy's Gram scalars as the step finds them
  if (GRAM && A.mode == MODE_STEP) {
    if (tid < 2 * LBM * LBM) { gr_old0 = A.gram[(size_t)dec * GR_N + tid]; s_gl.gram[tid] = gr_old0; }
    if (tid < 2 * LBM) { gr_old1 = A.gram[(size_t)dec * GR_N + 2 * LBM * LBM + tid]; s_gl.gram[2 * LBM * LBM + tid] = gr_old1; }
  }
  bool gram_dirty = false;
  bsync<NW>();
  const trx2_run* runs_l = reinterpret_cast<const trx2_run*>(s_runs);
  int run = s_i[SI_RUN], phase = s_i[SI_PHASE];
  if (A.mode == MODE_STEP && phase == PH_DONE) return;
  // the Cartesian role's turn -- except for the report, which is a torsion-space evaluation whatever the last run was
  if (A.mode == MODE_STEP && (s_i[SI_SEQ] == seq || (phase != PH_REPORT && runs_l[min(run, A.nruns - 1)].cartesian))) return;
  bool fresh_geom = false;  // a refilled slot starts from ideal bond geometry: use it without reading it back
  // internal geometry of residues 0, r and r + 1 for the NeRF pass, requested when the two-loop recursion starts (its ~5000
  // cycles cover the round trip; the recursion itself reads LDS only)
  float4 gpre[5];
  bool gpre_ok = false;

  const size_t vb = (size_t)dec * L;  // base of this decoy's [L] vectors
  float4 xt[RPT], gt[RPT];
  bool need_nerf = true;
  if (HIST_LDS && A.mode == MODE_STEP) {
    // the hl stored pairs, newest first; lanes beyond L copy the last residue (no branch around the load), never used
    // rows exactly L long (lanes beyond the chain are switched off: LDS-DMA writes only for active lanes): a 150-residue chain
    // stages 38 KB, not 64 -- what lets two step workgroups share a CU's LDS in the launches that have more of them than CUs
    const int hl0 = s_i[SI_HL], hh0 = s_i[SI_HH];
    if (tid < L)
      for (int kk = 0; kk < hl0; kk++) {
        const int j = (hh0 - 1 - kk + LBM) % LBM;
        lds_dma16(A.S + ((size_t)dec * LBM + j) * L + tid, s_hist + (j * 2 + 0) * L + wave * 64);
        lds_dma16(A.Y + ((size_t)dec * LBM + j) * L + tid, s_hist + (j * 2 + 1) * L + wave * 64);
      }
  }
  CSTAMP(0)  // state load, barrier, role test
  CCOUNT(30)

  if (A.mode != MODE_INIT) {
    // ------------------------------------------------------------------ consume the evaluation at XT
    const trx2_run R = runs_l[min(run, A.nruns - 1)];
    // accepted point, its gradient and the direction: needed only by the state machine below, loaded here so that their
    // latency overlaps the slab loads and the gradient assembly
    float4 x[RPT], g[RPT], dv[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int r = k * NT + tid;
      x[k] = g[k] = dv[k] = make_float4(0, 0, 0, 0);
      if (RPT == 1) { if (r < L) { x[k] = e_x; g[k] = e_g; dv[k] = e_dv; } }   // (zeros unless MODE_STEP)
      else if (r < L && A.mode == MODE_STEP) { x[k] = A.X[vb + r]; g[k] = A.G[vb + r]; dv[k] = A.D[vb + r]; }
    }
    double esum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    f3 g2[RPT], g1[RPT];       // per-residue sums of gradient / x cross gradient
    f3 gO_[RPT], gC_[RPT], gCB_[RPT], pN[RPT], pCA[RPT], pC[RPT], pO[RPT], pCB[RPT];
    float4 nxq[RPT];  // N of the next residue (the omega axis ends there): requested with this residue's atoms, used after the scan
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int r = k * NT + tid;
      xt[k] = make_float4(0, 0, 0, 0);
      nxq[k] = make_float4(0, 0, 0, 0);
      gt[k] = make_float4(0, 0, 0, 0);
      g2[k] = g1[k] = gO_[k] = gC_[k] = gCB_[k] = pN[k] = pCA[k] = pC[k] = pO[k] = pCB[k] = mk3(0, 0, 0);
      if (r < L) {
        const float4* xp = A.P + (vb + r) * 5;
        float4 c0, c1, c2, c3, c4;
        float g[PR_NCOMP], ep[6];
        if (RPT == 1) {
          xt[k] = e_xt; c0 = e_c[0]; c1 = e_c[1]; c2 = e_c[2]; c3 = e_c[3]; c4 = e_c[4]; nxq[k] = e_c[5];
          sum_pair_records<FA_NPRE>(A.FA, nsl_pre, A.B, L, dec, r, g, ep, e_fa);
        } else {
          xt[k] = A.XT[vb + r];
          c0 = xp[0]; c1 = xp[1]; c2 = xp[2]; c3 = xp[3]; c4 = xp[4];
          nxq[k] = xp[r + 1 < L ? 5 : 0];
          { const float4 none[1][6] = {}; sum_pair_records<0>(A.FA, (int)A.nslice[r], A.B, L, dec, r, g, ep, none); }
        }
        esum[0] += ep[0]; esum[1] += ep[1]; esum[2] += ep[2]; esum[3] += ep[3]; esum[4] += ep[4]; esum[8] += ep[5];
        pN[k] = mk3(c0.x, c0.y, c0.z); pCA[k] = mk3(c0.w, c1.x, c1.y); pC[k] = mk3(c1.z, c1.w, c2.x);
        pO[k] = mk3(c2.y, c2.z, c2.w); pCB[k] = mk3(c3.x, c3.y, c3.z);
        const f3 pH = mk3(c4.x, c4.y, c4.z), gH = mk3(g[15], g[16], g[17]);
        f3 gN = mk3(g[0], g[1], g[2]), gCA = mk3(g[3], g[4], g[5]);
        gC_[k] = mk3(g[6], g[7], g[8]); gO_[k] = mk3(g[9], g[10], g[11]); gCB_[k] = mk3(g[12], g[13], g[14]);
        // the backbone H rides on its residue: it is placed from C(r-1), N, CA, all upstream of this residue's own torsions,
        // so it belongs to the suffix sums (every torsion before r moves it) and to none of the residue's own moved sets
        g2[k] = gN + gCA + gC_[k] + gO_[k] + gCB_[k] + gH;
        g1[k] = cross(pN[k], gN) + cross(pCA[k], gCA) + cross(pC[k], gC_[k]) + cross(pO[k], gO_[k]) + cross(pCB[k], gCB_[k]) + cross(pH, gH);
        // torsion-space terms: rama (residues 2..L-1) and omega_bb (peptides 1..L-1; 2..L-1 with the fitted terms)
        if (r < L - 1) {
          float sph, cph, sps, cps;
          fast_sincosf(xt[k].x, &sph, &cph);
          fast_sincosf(xt[k].y, &sps, &cps);
          if (r >= 1) {
            float4 rp[3] = {e_rp[0], e_rp[1], e_rp[2]};
            if (RPT > 1 && TRX2_RAMA_FIT_ON) { const float4* rpp = rama_par_ptr(A.hasH, L) + (size_t)r * 3; rp[0] = rpp[0]; rp[1] = rpp[1]; rp[2] = rpp[2]; }
            float gph, gps;
            esum[5] += (double)rama_eval(sph, cph, sps, cps, rp, gph, gps);
            gt[k].x += R.w[4] * gph;
            gt[k].y += R.w[4] * gps;
          }
          if (omega_on(r, L)) {
            float gps, gom;
            esum[6] += (double)omega_eval(sps, cps, xt[k].z, gps, gom);
            gt[k].y += R.w[5] * gps;
            gt[k].z += R.w[5] * gom;
          }
        }
      }
    }
    if (HIST_LDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the wave: by now it has landed anyway
    CSTAMP(1)  // slab sums, coordinates, rama / omega per residue
    // ---- suffix sums over residues of (g2, g1): chunks from the end, wave shuffles + LDS wave totals
    f3 car2 = mk3(0, 0, 0), car1 = mk3(0, 0, 0);  // sum over all residues in later chunks
#pragma unroll
    for (int k = RPT - 1; k >= 0; k--) {
      float v[6] = {g2[k].x, g2[k].y, g2[k].z, g1[k].x, g1[k].y, g1[k].z};
      wave_suffix_sums<6>(v, lane);
      bsync<NW>();
      if (lane == 0)
#pragma unroll
        for (int i = 0; i < 6; i++) s_scan[wave * 6 + i] = v[i];
      bsync<NW>();
      float tot[6] = {0, 0, 0, 0, 0, 0}, after[6] = {0, 0, 0, 0, 0, 0};
      for (int w = 0; w < NW; w++)
#pragma unroll
        for (int i = 0; i < 6; i++) {
          float t = s_scan[w * 6 + i];
          tot[i] += t;
          if (w > wave) after[i] += t;
        }
      // inclusive suffix sum for this residue = v + later waves + later chunks; exclusive = minus own
      f3 inc2 = mk3(v[0] + after[0] + car2.x, v[1] + after[1] + car2.y, v[2] + after[2] + car2.z);
      f3 inc1 = mk3(v[3] + after[3] + car1.x, v[4] + after[4] + car1.y, v[5] + after[5] + car1.z);
      f3 ex2 = inc2 - g2[k], ex1 = inc1 - g1[k];  // residues > r
      const int r = k * NT + tid;
      if (r < L) {
        // omega_r: axis C_r -> N_{r+1}
        if (r + 1 < L) {
          const float4 nx = nxq[k];
          f3 Nn = mk3(nx.x, nx.y, nx.z);
          f3 n = unit(Nn - pC[k]);
          gt[k].z += dot(n, ex1) - dot(cross(n, pC[k]), ex2);
        }
        {  // psi_r: axis CA -> C, moves O_r and residues > r
          f3 h1 = ex1 + cross(pO[k], gO_[k]), h2 = ex2 + gO_[k];
          f3 n = unit(pC[k] - pCA[k]);
          gt[k].y += dot(n, h1) - dot(cross(n, pCA[k]), h2);
        }
        {  // phi_r: axis N -> CA, moves CB_r, C_r, O_r and residues > r
          f3 h1 = ex1 + cross(pO[k], gO_[k]) + cross(pC[k], gC_[k]) + cross(pCB[k], gCB_[k]);
          f3 h2 = ex2 + gO_[k] + gC_[k] + gCB_[k];
          f3 n = unit(pCA[k] - pN[k]);
          gt[k].x += dot(n, h1) - dot(cross(n, pN[k]), h2);
        }
      }
      car2 = car2 + mk3(tot[0], tot[1], tot[2]);
      car1 = car1 + mk3(tot[3], tot[4], tot[5]);
    }
    CSTAMP(2)  // suffix scan + torsion gradient
    // The nine terms are needed by the report, by the INIT / FINISH passes and by the guard of a pre-checked run; a plain
    // minimiser step needs their weighted total only: ONE f64 workgroup sum instead of nine.
    // Round 2 kept two residues per thread on nine sums: inside the fused kernel (k_step<2, 256, 512>, L = 400) the one-sum
    // build accepted no step in the declash runs.  The sums were never at fault (the checking build compares every one-sum
    // total with the nine-sum total: 0 mismatches): what the one-sum build lost was an assignment in the state machine below
    // (see `started`).  The totals are wave-uniform by construction and are marked so (uniform_d): every decision of the state
    // machine is a scalar branch.  tests/test_gpu_configs.py (configuration 4) and the checking build (TRX2_SELFCHECK,
    // tests/test_gpu_selfcheck.py: run starts whose window was not seeded) guard it.
    const bool all_terms = A.mode != MODE_STEP || phase == PH_REPORT || (phase == PH_START && (R.precheck & TRX2_RUN_PRECHECK));
    double f_t;
    // a line-search evaluation of the Gram form: the step this trial would make (s, y) and its products with the stored pairs are computed
    // BEFORE the Armijo test and reduced together with the energy (gram_reduce_with_energy)
    const bool fused = GRAM && A.mode == MODE_STEP && phase == PH_LS;
    float4 s_try[RPT], y_try[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) s_try[k] = y_try[k] = make_float4(0, 0, 0, 0);
    if (all_terms) {
      block_sum_n<9, NW>(esum, s_buf, flip);
      f_t = (double)R.w[0] * esum[0] + (double)R.w[1] * (esum[1] + esum[2]) + (double)R.w[2] * esum[3] +
            (double)R.w[3] * esum[4] + (double)R.w[4] * esum[5] + (double)R.w[5] * esum[6] + (double)R.w[7] * esum[8];
      if (tid == 0)  // one lane, constant indices: indexing the register array with the thread id moved it to scratch memory
#pragma unroll
        for (int k = 0; k < TRX2_NTERMS; k++) A.e_last[(size_t)dec * TRX2_NTERMS + k] = esum[k];
    } else {
      double ft1[1] = {(double)R.w[0] * esum[0] + (double)R.w[1] * (esum[1] + esum[2]) + (double)R.w[2] * esum[3] +
                       (double)R.w[3] * esum[4] + (double)R.w[4] * esum[5] + (double)R.w[5] * esum[6] + (double)R.w[7] * esum[8]};
      if constexpr (GRAM) {
        if (fused) {
          const int hl_ = s_i[SI_HL], hh_ = s_i[SI_HH];
#pragma unroll
          for (int k = 0; k < RPT; k++) {
            s_try[k] = make_float4(xt[k].x - x[k].x, xt[k].y - x[k].y, xt[k].z - x[k].z, 0);
            y_try[k] = make_float4(gt[k].x - g[k].x, gt[k].y - g[k].y, gt[k].z - g[k].z, 0);
          }
          // every product of the new pair and the new gradient with the stored pairs (age order), one fused reduction
          float pv[GV_N];
#pragma unroll
          for (int q = 0; q < GV_N; q++) pv[q] = 0.0f;
#pragma unroll
          for (int kr = 0; kr < RPT; kr++) {
            const int r = kr * NT + tid, rc = min(r, L - 1);
            // the reads of HB pairs issued together (a slot that holds no pair is read and discarded: one wait instead of HB).  Staged
            // history: all LBM pairs at once; from global memory (chains beyond 256 residues): four pairs at a time (registers)
            constexpr int HB = (HIST_LDS && NT <= 256) ? LBM : LBM / 2;   // (512 threads run at 256 registers: four pairs at a time)
#pragma unroll
            for (int k0 = 0; k0 < LBM; k0 += HB) {
              float4 so[HB], yo[HB];
#pragma unroll
              for (int k = 0; k < HB; k++) {
                const int j = (hh_ - 1 - (k0 + k) + LBM) % LBM;
                if (HIST_LDS) { so[k] = s_hist[(j * 2 + 0) * L + rc]; yo[k] = s_hist[(j * 2 + 1) * L + rc]; }
                else { so[k] = A.S[((size_t)dec * LBM + j) * L + rc]; yo[k] = A.Y[((size_t)dec * LBM + j) * L + rc]; }
              }
#pragma unroll
              for (int k = 0; k < HB; k++) {
                const bool on = k0 + k < hl_ && r < L;
                pv[GV_A + k0 + k] += on ? dot3(so[k], y_try[kr]) : 0.0f; pv[GV_B + k0 + k] += on ? dot3(yo[k], y_try[kr]) : 0.0f;
                if (k0 + k < LBM - 1) pv[GV_C + k0 + k] += on ? dot3(s_try[kr], yo[k]) : 0.0f;
              }
            }
            // (s, y, gt are zero beyond the chain)
            pv[GV_SY] += dot3(s_try[kr], y_try[kr]); pv[GV_SS] += dot3(s_try[kr], s_try[kr]); pv[GV_YY] += dot3(y_try[kr], y_try[kr]);
            pv[GV_SG] += dot3(s_try[kr], gt[kr]); pv[GV_YG] += dot3(y_try[kr], gt[kr]); pv[GV_GG] += dot3(gt[kr], gt[kr]);
          }
          f_t = gram_reduce_with_energy<GRAM ? NT : 16>(pv, ft1[0], s_gl, s_buf, flip);
        } else {
          block_sum_n<1, NW>(ft1, s_buf, flip);
          f_t = ft1[0];
        }
      } else {
        block_sum_n<1, NW>(ft1, s_buf, flip);
        f_t = ft1[0];
      }
    }
    f_t = uniform_d(f_t);
    if (tid == 0) A.f_last[dec] = f_t;
#ifdef TRX2_SELFCHECK
    // Checking build only (libtrx2fold_check.so, never the shipped library): wherever a step took the ONE-sum path, the
    // nine terms are reduced as well and their weighted total compared with it.  g_selfcheck[0] counts the checks, [1] the
    // mismatches (relative 1e-9: both are float64 sums of the same per-thread values in another order).
    if (!all_terms) {
      double chk[9];
#pragma unroll
      for (int k = 0; k < 9; k++) chk[k] = esum[k];
      block_sum_n<9, NW>(chk, s_buf, flip);
      const double f9 = (double)R.w[0] * chk[0] + (double)R.w[1] * (chk[1] + chk[2]) + (double)R.w[2] * chk[3] + (double)R.w[3] * chk[4] +
                        (double)R.w[4] * chk[5] + (double)R.w[5] * chk[6] + (double)R.w[7] * chk[8];
      if (tid == 0) {
        atomicAdd(&g_selfcheck[0], 1ull);
        if (!(fabs(f9 - f_t) <= 1e-9 * fabs(f9) + 1e-9)) atomicAdd(&g_selfcheck[1], 1ull);
      }
    }
#endif

    if (A.mode == MODE_STEP && phase == PH_REPORT) {
      // ---- this evaluation was the decoy's report: results out by decoy id, then the slot takes the next decoy or retires
      const int id = A.slot_id[dec];
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        const int r = k * NT + tid;
        if (r < L) {
          A.out_X[(size_t)id * L + r] = xt[k];
          const float4* xp = A.P + (vb + r) * 5;
          float4* xo = A.out_xyz + ((size_t)id * L + r) * 4;
          xo[0] = xp[0]; xo[1] = xp[1]; xo[2] = xp[2]; xo[3] = xp[3];
        }
      }
      if (tid == 0)
#pragma unroll
        for (int k = 0; k < TRX2_NTERMS; k++) A.out_e[(size_t)id * TRX2_NTERMS + k] = esum[k];
      __shared__ int s_new;
      if (tid == 0) {
        A.out_f[id] = f_t;
        int* st = A.out_stat + (size_t)id * 4;
        st[0] = s_i[SI_STATUS]; st[1] = s_i[SI_NEVALS]; st[2] = s_i[SI_NITERS]; st[3] = dec;
        s_new = atomicAdd(A.next_id, 1);
      }
      bsync<NW>();
      const int nid = s_new;
      if (nid >= A.n_total) {  // queue empty: the slot retires
        if (tid == 0) {
          gi[SI_PHASE] = PH_DONE;
          *reinterpret_cast<volatile unsigned long long*>(gi) = ((unsigned long long)(unsigned)seq << 32) | (unsigned long long)(unsigned)run;
          A.wcur[(size_t)dec * 8 + 6] = 0.0f;
          atomicAdd(A.done_count, 1);
        }
        return;
      }
      const ResGeom gid = ideal_geom();
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        const int r = k * NT + tid;
        if (r < L) {
          xt[k] = start_torsions(L, A.seed, A.decoy0 + (unsigned)nid, r, A.tors0_all ? A.tors0_all + (size_t)nid * L * 3 : nullptr);
          A.X[vb + r] = xt[k]; A.XT[vb + r] = xt[k];
          A.geom[(vb + r) * 3] = gid.g0; A.geom[(vb + r) * 3 + 1] = gid.g1; A.geom[(vb + r) * 3 + 2] = gid.g2;
        } else xt[k] = make_float4(0, 0, 0, 0);
      }
      run = 0; phase = PH_START; fresh_geom = true; need_nerf = true;
      if (tid == 0) {
        A.slot_id[dec] = nid;
        gi[SI_PHASE] = PH_START; gi[SI_ITER] = 0; gi[SI_NLS] = 0; gi[SI_HL] = 0; gi[SI_HH] = 0; gi[SI_NH] = 0; gi[SI_STATUS] = 0;
        gi[SI_NEVALS] = 0; gi[SI_NITERS] = 0; gi[SI_WSPACE] = 0;
        *reinterpret_cast<volatile unsigned long long*>(gi) = ((unsigned long long)(unsigned)seq << 32);  // run 0, stepped in this launch
        for (int q = 0; q < SD_N; q++) gd_[q] = 0.0;
      }
      goto next_pair;
    }

    if (A.mode == MODE_FINISH) {
      if (A.grad_out)
#pragma unroll
        for (int k = 0; k < RPT; k++) {
          const int r = k * NT + tid;
          if (r < L) {
            float* go = A.grad_out + (vb + r) * 3;
            go[0] = gt[k].x; go[1] = gt[k].y; go[2] = gt[k].z;
          }
        }
      return;
    }

    // ------------------------------------------------------------------ minimiser state machine (uniform)
    int iter = s_i[SI_ITER], nls = s_i[SI_NLS], hl = s_i[SI_HL], hh = s_i[SI_HH], nh = s_i[SI_NH];
    int n_evals = s_i[SI_NEVALS] + 1, n_iters = s_i[SI_NITERS], status = s_i[SI_STATUS];
    double f = s_d[SD_F], alpha = s_d[SD_ALPHA], gdir = s_d[SD_GD];
    double fh[3] = {s_d[SD_FH0], s_d[SD_FH1], s_d[SD_FH2]};
    double gamma_h = s_d[SD_GAMMA];
    int wspace = s_i[SI_WSPACE];
    bool next_run = false, new_dir = false, steepest = false, new_trial = false, started = false;
    float4 s_new[RPT], y_new[RPT];  // two residues per thread, Gram form: the pair stored in this step, kept for the direction pass
    bool have_new = false;
#pragma unroll
    for (int k = 0; k < RPT; k++) s_new[k] = y_new[k] = make_float4(0, 0, 0, 0);
    const bool finite_t = isfinite(f_t);
    CSTAMP(3)  // energy reduction + loads of X, G, D
    if (!finite_t && phase == PH_START) { status = TRX2_DIVERGED; phase = PH_DONE; }
    else if (phase == PH_START) {
      // (a pre-checked start reduced all nine terms: esum[] holds workgroup totals here)
      if ((R.precheck & TRX2_RUN_PRECHECK) && uniform_d(esum[5]) + (double)TRX2_RAMA_GUARD_OFFSET * (double)max(L - 2, 0) + uniform_d(esum[4]) < (double)TRX2_CLASH_BREAK) {
        run = R.skip_to;
        if (run >= A.nruns) phase = PH_DONE;
        need_nerf = false;
      } else {
        f = f_t;
#pragma unroll
        for (int k = 0; k < RPT; k++) g[k] = gt[k];
        started = true;   // history and non-monotone window reset below, on the path every arm joins
        steepest = true;
      }
    } else {  // PH_LS
      double fref = fh[0];
      for (int k = 1; k < nh; k++) fref = fmax(fref, fh[k]);
      const bool accept = finite_t && f_t <= fref + (double)TRX2_LS_C1 * alpha * gdir;
      if (accept) {
        double v3[3] = {0, 0, 0};
        float4 s[RPT], y[RPT];
#pragma unroll
        for (int k = 0; k < RPT; k++) {
          s[k] = make_float4(xt[k].x - x[k].x, xt[k].y - x[k].y, xt[k].z - x[k].z, 0);
          y[k] = make_float4(gt[k].x - g[k].x, gt[k].y - g[k].y, gt[k].z - g[k].z, 0);
        }
        if constexpr (GRAM) {
          // the products came with the energy (`fused`: every PH_LS evaluation of this form is)
          v3[0] = s_gl.out[GV_SY]; v3[1] = s_gl.out[GV_SS]; v3[2] = s_gl.out[GV_YY];
        } else {
#pragma unroll
          for (int k = 0; k < RPT; k++) { v3[0] += (double)dot3(s[k], y[k]); v3[1] += (double)dot3(s[k], s[k]); v3[2] += (double)dot3(y[k], y[k]); }
          block_sum_n<3, NW>(v3, s_buf, flip);
        }
        const bool store = v3[0] > 1e-12 * sqrt(v3[1] * v3[2]);
        if (store) {
#pragma unroll
          for (int k = 0; k < RPT; k++) {
            const int r = k * NT + tid;
            if (r < L) {
              A.S[((size_t)dec * LBM + hh) * L + r] = s[k];
              A.Y[((size_t)dec * LBM + hh) * L + r] = y[k];
            }
          }
          if (HIST_LDS) {  // slot hh of the staged copy (the oldest pair: its products are taken)
            if (tid < L) {
              s_hist[(hh * 2 + 0) * L + tid] = s[0];
              s_hist[(hh * 2 + 1) * L + tid] = y[0];
            }
          } else if (!GRAM) {
            bsync<NW>();
            if (tid == 0) s_rho[hh] = (float)(1.0 / v3[0]);
          }
          gamma_h = v3[0] / v3[2];
          wspace = 1;
          if (!GRAM) bsync<NW>();
          if (GRAM && !HIST_LDS) {  // the new pair, for the direction pass below (its global copy was stored a moment ago)
#pragma unroll
            for (int k = 0; k < RPT; k++) { s_new[k] = s[k]; y_new[k] = y[k]; }
            have_new = true;
          }
          hh = (hh + 1) % LBM;
          if (hl < LBM) hl++;
        }
        if constexpr (GRAM) {
          gram_advance<GRAM ? NT : 16>(s_gl, store, gr_old0, gr_old1);  // ends with a barrier: the LDS copy of the pair is visible too
          gram_dirty = true;
        }
        const double fprev = f;
#pragma unroll
        for (int k = 0; k < RPT; k++) { x[k] = xt[k]; g[k] = gt[k]; }
        f = f_t;
        if (nh < TRX2_LS_PAST) fh[nh++] = f;
        else { fh[0] = fh[1]; fh[1] = fh[2]; fh[2] = f; }
        iter++; n_iters++;
        const bool conv = 2.0 * fabs(fprev - f) <= (R.tol > 0.0f ? (double)R.tol : (double)TRX2_MIN_TOL) * (fabs(fprev) + fabs(f) + 1e-10);
        if (conv || iter >= R.max_iter) next_run = true;
        else new_dir = true;
      } else {
        nls++;
        alpha *= (double)TRX2_LS_SHRINK;
        if (nls > TRX2_LS_MAXTRIAL) {
          if (hl > 0) { hl = 0; steepest = true; }
          else next_run = true;
        } else new_trial = true;
      }
    }
    // A run that has just started: empty history, the window holds its first energy.  Done HERE, after the arms have joined,
    // and not inside the arm that starts the run: with the one-sum energy path in the same kernel, ROCm 7.2's hipcc emitted
    // the pre-checked start (`precheck` set, guard not met) without the `fh[0] = f` that its source arm contained -- the
    // assignment survived only in the arm of runs without a precheck (ISA of k_step<2, 256, 512>, both with divergent and with
    // scalar branches; profiles/README.md round 3).  fh[0] then kept the previous decoy's or run's value (0 after a reset),
    // every trial failed the Armijo test against it, and the declash runs made no progress: round 2's "never accepts a step".
    if (started) { hl = 0; hh = 0; nh = 1; fh[0] = f; iter = 0; }
#ifdef TRX2_SELFCHECK
    // a run that has just started (steepest-descent restart from PH_START) must have put its energy into the non-monotone window
    if (tid == 0 && s_i[SI_PHASE] == PH_START && steepest) {
      atomicAdd(&g_selfcheck[4], 1ull);
      if (!(fh[0] == f && nh == 1)) atomicAdd(&g_selfcheck[5], 1ull);
    }
#endif
#ifdef TRX2_DBG
    if (dec == 0 && tid == 0 && A.mode == MODE_STEP && n_evals < 256) {
      double* q = g_dbg[n_evals];
      q[0] = s_i[SI_PHASE]; q[1] = s_i[SI_RUN]; q[2] = f_t; q[3] = s_d[SD_F]; q[4] = s_d[SD_ALPHA]; q[5] = s_d[SD_GD];
      q[6] = (double)new_dir + 2.0 * (double)steepest + 4.0 * (double)next_run + 8.0 * (double)new_trial; q[7] = s_i[SI_NLS]; q[8] = s_i[SI_HL];
      q[9] = fh[0]; q[10] = (double)xt[0].x; q[11] = (double)gt[0].x;
    }
#endif
    CSTAMP(4)  // Armijo test; on acceptance the (s, y) pair: one reduction + stores
    if (new_dir) {
      CCOUNT(31)
      // Two-loop recursion over the stored pairs (A.S / A.Y are L2-resident; coalesced float4 per residue): 2 x hl dependent
      // rounds of { dot with the pair, reduce, axpy }, ~1000 cycles each = 40 % of a step (s_memtime stamps,
      // profiles/README.md).  The next pair is requested before the current reduction.  Measured and NOT kept: the whole
      // recursion on one wave (in-wave sums only, no barriers) and two-deep prefetch through three rotating register
      // buffers -- all within 2 % of this form: a round is bound by its own serial arithmetic (f64 DPP sum, readlanes, the
      // LDS read of rho, the axpy), not by the loads or the barriers.
      double v2[2] = {0, 0};
      auto pair_at = [&](int kk) { return (hh - 1 - kk + LBM) % LBM; };
      if constexpr (GRAM) {
        if constexpr (HIST_LDS) {
          // internal geometry for the NeRF pass: requested here, used after the direction is built
          const float4* gq0 = A.geom + vb * 3;
          const int rq = min(tid, L - 1), rn = min(tid + 1, L - 1);
          gpre[0] = gq0[0]; gpre[1] = gq0[rq * 3]; gpre[2] = gq0[rq * 3 + 1]; gpre[3] = gq0[rq * 3 + 2]; gpre[4] = gq0[rn * 3];
          gpre_ok = true;
        }
        float cy[LBM], cs[LBM];
        double g_r;
        gram_recursion(s_gl.gram, hl, gamma_h, s_gl.out[GV_GG], cy, cs, g_r);
        CSTAMP(5)  // recurrences on the Gram scalars
        const float gam = (float)gamma_h;
#pragma unroll
        for (int kr = 0; kr < RPT; kr++) {
          const int r = kr * NT + tid, rc = min(r, L - 1);
          float4 q = make_float4(gam * g[kr].x, gam * g[kr].y, gam * g[kr].z, 0);
          constexpr int HB = (HIST_LDS && NT <= 256) ? LBM : LBM / 2;
#pragma unroll
          for (int m0 = 0; m0 < LBM; m0 += HB) {
            float4 sm[HB], ym[HB];
#pragma unroll
            for (int m = 0; m < HB; m++) {
              const int j = pair_at(m0 + m);
              if (HIST_LDS) { sm[m] = s_hist[(j * 2 + 0) * L + rc]; ym[m] = s_hist[(j * 2 + 1) * L + rc]; }
              else if (m0 + m == 0 && have_new) { sm[0] = s_new[kr]; ym[0] = y_new[kr]; }   // stored a moment ago: from registers
              else { sm[m] = A.S[((size_t)dec * LBM + j) * L + rc]; ym[m] = A.Y[((size_t)dec * LBM + j) * L + rc]; }
            }
#pragma unroll
            for (int m = 0; m < HB; m++) {
              const float a = m0 + m < hl ? cs[m0 + m] : 0.0f, b = m0 + m < hl ? cy[m0 + m] : 0.0f;
              const float4 z = make_float4(0, 0, 0, 0), s_ = m0 + m < hl ? sm[m] : z, y_ = m0 + m < hl ? ym[m] : z;  // an unused slot may hold anything
              q.x = fmaf(b, y_.x, fmaf(a, s_.x, q.x)); q.y = fmaf(b, y_.y, fmaf(a, s_.y, q.y)); q.z = fmaf(b, y_.z, fmaf(a, s_.z, q.z));
            }
          }
          dv[kr] = r < L ? make_float4(-q.x, -q.y, -q.z, 0) : make_float4(0, 0, 0, 0);
        }
        v2[0] = -g_r; v2[1] = s_gl.out[GV_GG];
        CSTAMP(7)  // direction from the stored vectors
      } else
      {
        float4 q[RPT];
#pragma unroll
        for (int k = 0; k < RPT; k++) q[k] = g[k];
        auto load_pair = [&](int j, float4 (&s_)[RPT], float4 (&y_)[RPT]) {
          if (HIST_LDS) {
            const float4 z = make_float4(0, 0, 0, 0), sv = s_hist[(j * 2 + 0) * L + min(tid, L - 1)], yv = s_hist[(j * 2 + 1) * L + min(tid, L - 1)];
            s_[0] = tid < L ? sv : z;
            y_[0] = tid < L ? yv : z;
            return;
          }
#pragma unroll
          for (int k = 0; k < RPT; k++) {
            const int r = k * NT + tid;
            s_[k] = y_[k] = make_float4(0, 0, 0, 0);
            if (r < L) { s_[k] = A.S[((size_t)dec * LBM + j) * L + r]; y_[k] = A.Y[((size_t)dec * LBM + j) * L + r]; }
          }
        };
        float4 sj[RPT], yj[RPT], sn[RPT], yn[RPT];
        if (hl > 0) load_pair(pair_at(0), sj, yj);
        for (int kk = 0; kk < hl; kk++) {
          const int j = pair_at(kk);
          load_pair(pair_at(kk + 1 < hl ? kk + 1 : hl - 1), sn, yn);
          double v1[1] = {0};
#pragma unroll
          for (int k = 0; k < RPT; k++) v1[0] += (double)dot3(sj[k], q[k]);
          block_sum_n<1, NW>(v1, s_buf, flip);
          const float al = s_rho[j] * (float)v1[0];
          if (tid == 0) s_alpha[j] = al;
#pragma unroll
          for (int k = 0; k < RPT; k++) {
            q[k].x -= al * yj[k].x; q[k].y -= al * yj[k].y; q[k].z -= al * yj[k].z;
            sj[k] = sn[k]; yj[k] = yn[k];
          }
        }
        CSTAMP(5)  // two-loop, first loop
        if (hl > 0) {
          const float gam = (float)gamma_h;
#pragma unroll
          for (int k = 0; k < RPT; k++) { q[k].x *= gam; q[k].y *= gam; q[k].z *= gam; }
        }
        bsync<NW>();
        for (int kk = hl - 1; kk >= 0; kk--) {
          const int j = pair_at(kk);
          if (kk > 0) load_pair(pair_at(kk - 1), sn, yn);
          double v1[1] = {0};
#pragma unroll
          for (int k = 0; k < RPT; k++) v1[0] += (double)dot3(yj[k], q[k]);
          block_sum_n<1, NW>(v1, s_buf, flip);
          const float c = s_alpha[j] - s_rho[j] * (float)v1[0];
#pragma unroll
          for (int k = 0; k < RPT; k++) {
            q[k].x += c * sj[k].x; q[k].y += c * sj[k].y; q[k].z += c * sj[k].z;
            sj[k] = sn[k]; yj[k] = yn[k];
          }
        }
        CSTAMP(7)  // two-loop, second loop
#pragma unroll
        for (int k = 0; k < RPT; k++) {
          dv[k] = make_float4(-q[k].x, -q[k].y, -q[k].z, 0);
          v2[0] += (double)dot3(g[k], dv[k]); v2[1] += (double)dot3(g[k], g[k]);
        }
        block_sum_n<2, NW>(v2, s_buf, flip);
      }
      if (!(v2[1] > 0)) next_run = true;
      else if (hl == 0 || !(v2[0] < 0)) { hl = 0; steepest = true; }
      else { gdir = v2[0]; alpha = 1.0; nls = 0; new_trial = true; }
    }
    if (steepest) {
      double v1[1] = {0};
#pragma unroll
      for (int k = 0; k < RPT; k++) { dv[k] = make_float4(-g[k].x, -g[k].y, -g[k].z, 0); v1[0] += (double)dot3(g[k], g[k]); }
      block_sum_n<1, NW>(v1, s_buf, flip);
      if (!(v1[0] > 0)) next_run = true;
      else {
        gdir = -v1[0];
        // a run's first step: the Hessian scale of the last pair stored in this space (the previous run's), not longer than a unit step
        alpha = ((R.precheck & TRX2_RUN_WARM) && started && wspace == 1 && gamma_h > 0.0) ? fmin(gamma_h, 1.0 / sqrt(v1[0])) : fmin(1.0, 1.0 / sqrt(v1[0]));
        nls = 0;
        new_trial = true;
      }
    }
    CSTAMP(8)  // descent test / steepest-descent restart
    if (next_run) {
      run++;
      phase = (run >= A.nruns) ? PH_DONE : PH_START;
#pragma unroll
      for (int k = 0; k < RPT; k++) xt[k] = x[k];
      need_nerf = true;  // XT := X (the accepted point) so that coordinates match for the next evaluation
    }
    if (new_trial) {
      phase = PH_LS;
      const float al = (float)alpha;
#pragma unroll
      for (int k = 0; k < RPT; k++)
        xt[k] = make_float4(fmaf(al, dv[k].x, x[k].x), fmaf(al, dv[k].y, x[k].y), fmaf(al, dv[k].z, x[k].z), 0);
    }
    if (phase != PH_DONE && n_evals >= A.max_evals) { status = TRX2_MAXEVAL; phase = PH_DONE; }
    if (phase == PH_DONE) {  // over (protocol finished, budget spent, or diverged): next comes the report at the accepted point
      phase = PH_REPORT; run = A.nruns - 1;
#pragma unroll
      for (int k = 0; k < RPT; k++) xt[k] = x[k];
      need_nerf = true;
    }
    // ---- store state
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int r = k * NT + tid;
      if (r < L) { A.X[vb + r] = x[k]; A.G[vb + r] = g[k]; A.D[vb + r] = dv[k]; A.XT[vb + r] = xt[k]; }
    }
    bsync<NW>();
    if (GRAM && A.mode == MODE_STEP && (gram_dirty || hl == 0)) {
      // the Gram scalars follow the history; a restarted history starts from zeros (entries of absent pairs are only ever
      // multiplied by zero coefficients, so they must stay finite)
      const bool z = hl == 0;
      if (tid < 2 * LBM * LBM) A.gram[(size_t)dec * GR_N + tid] = z ? 0.0 : s_gl.gram[tid];
      if (tid < 2 * LBM) A.gram[(size_t)dec * GR_N + 2 * LBM * LBM + tid] = z ? 0.0 : s_gl.gram[2 * LBM * LBM + tid];
    }
    if (tid == 0) {
      gi[SI_PHASE] = phase; gi[SI_ITER] = iter; gi[SI_NLS] = nls; gi[SI_HL] = hl; gi[SI_HH] = hh;
      gi[SI_NH] = nh; gi[SI_STATUS] = status; gi[SI_NEVALS] = n_evals; gi[SI_NITERS] = n_iters; gi[SI_WSPACE] = wspace;
      *reinterpret_cast<volatile unsigned long long*>(gi) = ((unsigned long long)(unsigned)seq << 32) | (unsigned long long)(unsigned)run;  // last, in one piece
      gd_[SD_F] = f; gd_[SD_ALPHA] = alpha; gd_[SD_GD] = gdir; gd_[SD_FH0] = fh[0]; gd_[SD_FH1] = fh[1]; gd_[SD_FH2] = fh[2];
      gd_[SD_GAMMA] = gamma_h;
    }
    if (tid < LBM) A.rho[(size_t)dec * LBM + tid] = s_rho[tid];
  } else {
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int r = k * NT + tid;
      xt[k] = (r < L) ? A.XT[vb + r] : make_float4(0, 0, 0, 0);
    }
  }

  CSTAMP(9)  // trial point, state stores
next_pair:
  // ------------------------------------------------------------------ weights for the next pair launch
  if (tid == 0) {
    const trx2_run Rn = runs_l[min(run, A.nruns - 1)];
    float* w = A.wcur + (size_t)dec * 8;
    w[0] = Rn.w[0]; w[1] = Rn.w[1]; w[2] = Rn.w[2]; w[3] = Rn.w[3];
    w[4] = (float)Rn.sep_lo; w[5] = (float)Rn.sep_hi;
    w[6] = (phase == PH_DONE && A.mode == MODE_STEP) ? 0.0f : 1.0f + (float)Rn.pair_filter;  // the pair kernel's selection: 1 + TRX2_FILTER_*
    w[7] = Rn.w[7];
  }
  if (!need_nerf) return;

  // ------------------------------------------------------------------ K1: torsions XT -> backbone (NeRF scan)
  // M_r maps frame r+1 coordinates into frame r; F_r = F_0 o M_0 o ... o M_{r-1}
  bsync<NW>();
#pragma unroll
  for (int k = 0; k < RPT; k++) s_phi[k * NT + tid] = xt[k].x;
  bsync<NW>();
  const float4* gq = A.geom + vb * 3;
  const ResGeom g_ideal = ideal_geom();
  Xf carry;  // F_0: N at the origin, CA on +x, C in the xy plane (same start as the oracle)
  {
    const float4 q0 = fresh_geom ? g_ideal.g0 : (RPT == 1 && gpre_ok) ? gpre[0] : gq[0];
    float sa, ca;
    fast_sincosf(q0.w, &sa, &ca);
    carry = xf_from_atoms(mk3(0, 0, 0), mk3(q0.x, 0, 0), mk3(q0.x - q0.y * ca, q0.y * sa, 0));
  }
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int r = k * NT + tid;
    Xf M = xf_identity();
    ResGeom gr = ideal_geom();
    f3 lN = mk3(0, 0, 0), lCA = lN, lC = lN, lCB = lN, lNn = lN, lCAn = lN;
    if (r < L) {
      if (RPT == 1 && gpre_ok && !fresh_geom) { gr.g0 = gpre[1]; gr.g1 = gpre[2]; gr.g2 = gpre[3]; }
      else if (!fresh_geom) { gr.g0 = gq[r * 3]; gr.g1 = gq[r * 3 + 1]; gr.g2 = gq[r * 3 + 2]; }
      local_atoms(gr, lN, lCA, lC, lCB);
      if (r + 1 < L) {
        const float4 n0 = fresh_geom ? g_ideal.g0 : (RPT == 1 && gpre_ok) ? gpre[4] : gq[(r + 1) * 3];  // next residue: |N-CA|, |CA-C|, angle N-CA-C
        float spsi, cpsi, so, co, sp, cp, s1, c1, s2, c2, s3, c3;
        fast_sincosf(xt[k].y, &spsi, &cpsi);
        fast_sincosf(xt[k].z, &so, &co);
        fast_sincosf(s_phi[r + 1], &sp, &cp);  // phi of residue r+1
        fast_sincosf(gr.g1.x, &s1, &c1);
        fast_sincosf(gr.g1.y, &s2, &c2);
        fast_sincosf(n0.w, &s3, &c3);
        f3 Nn = place_atom(lN, lCA, lC, gr.g0.z, c1, s1, cpsi, spsi);
        f3 CAn = place_atom(lCA, lC, Nn, n0.x, c2, s2, co, so);
        f3 Cn = place_atom(lC, Nn, CAn, n0.y, c3, s3, cp, sp);
        M = xf_from_atoms(Nn, CAn, Cn);
        lNn = Nn; lCAn = CAn;
      }
    }
    CSTAMP(10)  // NeRF: geometry loads, sincos, local frames
    // inclusive scan of M within the wave
    Xf P = xf_wave_scan(M, lane);
    bsync<NW>();
    if (lane == 63) {
#pragma unroll
      for (int i = 0; i < 9; i++) s_scan[wave * 12 + i] = P.r[i];
#pragma unroll
      for (int i = 0; i < 3; i++) s_scan[wave * 12 + 9 + i] = P.t[i];
    }
    bsync<NW>();
    Xf pre = carry;  // transform of everything before this wave
    Xf tot = carry;
    for (int w = 0; w < NW; w++) {
      Xf T;
#pragma unroll
      for (int i = 0; i < 9; i++) T.r[i] = s_scan[w * 12 + i];
#pragma unroll
      for (int i = 0; i < 3; i++) T.t[i] = s_scan[w * 12 + 9 + i];
      if (w < wave) pre = xf_compose(pre, T);
      tot = xf_compose(tot, T);
    }
    CSTAMP(11)  // NeRF: scan of rigid transforms (in-wave + across waves)
    // frame of residue r = pre o (inclusive scan of the previous lane)
    Xf prev = xf_from_lane_below(P);
    Xf F = (lane == 0) ? pre : xf_compose(pre, prev);
    carry = tot;
    if (r < L) {
      float so_, co_, s4, c4;
      fast_sincosf(xt[k].y + gr.g2.x, &so_, &co_);  // dihedral N-CA-C-O = psi + t_O (ideal: pi)
      fast_sincosf(gr.g1.w, &s4, &c4);
      f3 lO = place_atom(lN, lCA, lC, gr.g1.z, c4, s4, co_, so_);
      f3 N = xf_apply(F, lN), CA = xf_apply(F, lCA), C = xf_apply(F, lC), O = xf_apply(F, lO), CB = xf_apply(F, lCB);
      float4 o0 = make_float4(N.x, N.y, N.z, CA.x), o1 = make_float4(CA.y, CA.z, C.x, C.y),
             o2 = make_float4(C.z, O.x, O.y, O.z), o3 = make_float4(CB.x, CB.y, CB.z, 0);
      float4* xo = A.P + (vb + r) * 5;
      xo[0] = o0; xo[1] = o1; xo[2] = o2; xo[3] = o3;
      // backbone H of the NEXT residue: its three parents C(r), N(r+1), CA(r+1) are all known in this residue's frame
      float4* xT = nullptr;
      if (A.xyzT) {
        xT = A.xyzT + ((size_t)((dec / A.BW) * L + r) * 5) * A.BW + dec % A.BW;
        float4 t0, t1, t2, t3;
        xt_pack(N, CA, C, O, CB, t0, t1, t2, t3);
        xT[0] = t0; xT[A.BW] = t1; xT[2 * A.BW] = t2; xT[3 * A.BW] = t3;
      }
      if (r + 1 < L) {
        const f3 Hn = xf_apply(F, place_h(lC, lNn, lCAn));
        const float4 hv = make_float4(Hn.x, Hn.y, Hn.z, (RPT == 1 ? hH_next : A.hasH[r + 1] != 0) ? 1.0f : 0.0f);
        xo[5 + 4] = hv;
        if (xT) xT[(5 + 4) * A.BW] = hv;
      }
      if (r == 0) {  // residue 1 has no predecessor: no H, a harmless position
        const float4 hv = make_float4(N.x + 1.0f, N.y, N.z, 0.0f);
        xo[4] = hv;
        if (xT) xT[4 * A.BW] = hv;
      }
    }
    CSTAMP(12)  // NeRF: atoms from frames, coordinate stores
  }
}

// =================================================================================================
// Cartesian-space minimiser step (MinMover.cartesian(True) on sf_cart, folding.py:83-84,100-102).  One workgroup per
// decoy, one residue per thread (L <= 256).  DOFs = the 15 coordinates of a residue, stored as 4 float4 (16th = 0):
// the trial vector IS the xyz buffer.  The pair kernel's gradient slabs are already Cartesian; added here: rama and
// omega from coordinates, and the harmonic bonded term (cart_bonded surrogate, trx2_model.h).  Terms that span two
// residues are evaluated by both owners, each keeping the gradient on its own atoms (no atomics).  When the run ends
// the relaxed internal geometry is extracted so that later torsion-space runs continue from it (oracle:
// orc_extract_internal).  The L-BFGS state machine is the one of k_chain on 4 float4 per residue.
// =================================================================================================
struct CartArgs {
  int L, B, nruns, max_evals;
  const int* seq_ctr;
  const trx2_run* runs;
  int* st_i; double* st_d; float* rho; double* gram;
  float4 *CX, *CG, *CD;      // [B][L][4] accepted point, its gradient, direction
  float4 *CS, *CY;           // [B][LBM][4][L]: stored pairs, component-major so that a wave reads 1 KB in one piece
  int hist_lds;              // stored pairs the launch has dynamic LDS for in this role (CART_HIST_BYTES each); 0: none
  float4* P;                 // [B][L][5]: the first 4 float4 of a record = trial coordinates = trial DOF vector (in/out), 5th = H
  float4* xyzT; int BW;      // decoy-minor copy for the pair kernel
  float4 *X, *XT, *geom;     // torsions and internal geometry, written when the run ends
  float* wcur;
  const float* FA; const unsigned char* nslice; int ns_max;
  const unsigned char* hasH;
  double *e_last, *f_last;
  int* done_count;
};
__device__ __forceinline__ float dot4(float4 a, float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }
struct Res5 { f3 N, CA, C, O, CB; };
template <typename P>
__device__ __forceinline__ Res5 unpack5(P p) {
  return Res5{mk3(p[0], p[1], p[2]), mk3(p[3], p[4], p[5]), mk3(p[6], p[7], p[8]), mk3(p[9], p[10], p[11]), mk3(p[12], p[13], p[14])};
}
__device__ __forceinline__ float wrap_pi_f(float x) { return x - 2.0f * TRX2_PI_F * rintf(x * (0.5f / TRX2_PI_F)); }
// harmonic bond: energy, gradient on a (gradient on b is the negative)
__device__ __forceinline__ float hbond(f3 a, f3 b, float d0, float k, f3& ga) {
  f3 u = a - b; float d = sqrtf(dot(u, u)), dd = d - d0; ga = u * (2.0f * k * dd / d); return k * dd * dd;
}
__device__ __forceinline__ float hangle(f3 a, f3 b, f3 c, float a0, float k, f3& ga, f3& gb, f3& gc) {
  float x = angle_grad(a, b, c, ga, gb, gc), dx = x - a0, sc = 2.0f * k * dx; ga = ga * sc; gb = gb * sc; gc = gc * sc; return k * dx * dx;
}
__device__ __forceinline__ float hdih(f3 a, f3 b, f3 c, f3 d, float t0, float k, f3& ga, f3& gb, f3& gc, f3& gd) {
  float x = dihedral_grad(a, b, c, d, ga, gb, gc, gd), dx = wrap_pi_f(x - t0), sc = 2.0f * k * dx;
  ga = ga * sc; gb = gb * sc; gc = gc * sc; gd = gd * sc; return k * dx * dx;
}
// link terms of the peptide bond P (residue i) -> Q (residue i+1): bond C-N', angles CA-C-N', C-N'-CA', O-C-N', improper CA-N'-C-O
struct LinkGrad { f3 CA, C, O, Nn, CAn; float e; };
__device__ __forceinline__ LinkGrad link_terms(const Res5& P, const Res5& Q) {
  LinkGrad G; G.CA = G.C = G.O = G.Nn = G.CAn = mk3(0, 0, 0);
  const float KL = (float)TRX2_CART_KLEN, KA = (float)TRX2_CART_KANG, KI = (float)TRX2_CART_KIMP;
  f3 a, b, c, d;
  float e = hbond(P.C, Q.N, (float)TRX2_B_C_N, KL, a); G.C += a; G.Nn += a * -1.0f;
  e += hangle(P.CA, P.C, Q.N, (float)TRX2_A_CA_C_N * TRX2_DEG_F, KA, a, b, c); G.CA += a; G.C += b; G.Nn += c;
  e += hangle(P.C, Q.N, Q.CA, (float)TRX2_A_C_N_CA * TRX2_DEG_F, KA, a, b, c); G.C += a; G.Nn += b; G.CAn += c;
  e += hangle(P.O, P.C, Q.N, 2.0f * TRX2_PI_F - (float)(TRX2_A_CA_C_N + TRX2_A_CA_C_O) * TRX2_DEG_F, KA, a, b, c); G.O += a; G.C += b; G.Nn += c;
  e += hdih(P.CA, Q.N, P.C, P.O, TRX2_PI_F, KI, a, b, c, d); G.CA += a; G.Nn += b; G.C += c; G.O += d;
  G.e = e; return G;
}

// One stored pair of the Cartesian role for this thread's residue: from its staged LDS slot, or from global memory (pair j).
// (`hist` typed as an LDS pointer: as a plain pointer the compiler merged the two sources into one FLAT load of a selected
// address -- the LDS reads of the staged pairs went through the flat path.)
typedef float v4f_t __attribute__((ext_vector_type(4)));   // (a builtin vector: HIP's float4 class has no assignment across address spaces)
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) v4f_t lds_f4;
__device__ __forceinline__ float4 lds_get(const lds_f4* p) { const v4f_t t = *p; return make_float4(t.x, t.y, t.z, t.w); }
__device__ __forceinline__ void lds_put(lds_f4* p, float4 v) { const v4f_t t = {v.x, v.y, v.z, v.w}; *p = t; }
template <int NT>
__device__ __forceinline__ void cart_hist_fetch(const CartArgs& A, const lds_f4* hist, int dec, int L, int rc, bool staged, int slot, int j, float4 (&s_)[4], float4 (&y_)[4]) {
  if (NT <= 256 && staged) {
#pragma unroll
    for (int q = 0; q < 4; q++) { s_[q] = lds_get(hist + (size_t)((slot * 2 + 0) * 4 + q) * L + rc); y_[q] = lds_get(hist + (size_t)((slot * 2 + 1) * 4 + q) * L + rc); }
  } else {
    const size_t o = ((size_t)dec * LBM + j) * 4 * L + rc;
#pragma unroll
    for (int q = 0; q < 4; q++) { s_[q] = A.CS[o + (size_t)q * L]; y_[q] = A.CY[o + (size_t)q * L]; }
  }
}
// LOWREG: the instantiation for launches with more step workgroups than the chip has CUs (more than 256 slots): the same
// arithmetic in the same order, but one stored pair at a time instead of two named buffers, so that the fused kernel fits 256
// registers and TWO workgroups run on a CU (one wave per SIMD otherwise: 256 + 106 registers).
template <int NT, bool LOWREG>
__device__ __forceinline__ void cart_body(const CartArgs& A, const int dec, int* s_runs, GramLds<NT>& s_gl) {
  constexpr int NW = NT / 64;  // one residue per thread: NT = 256 for chains up to 256 residues, 512 up to 512
  const int L = A.L, tid = threadIdx.x, r = tid;
  const bool act = r < L;
  __shared__ double s_buf[2 * NW * SBUF_K];
  int flip = 0;
  __shared__ float s_alpha[LBM];
  __shared__ int s_i[SI_N];
  __shared__ double s_d[SD_N];
  __shared__ float s_rho[LBM];
  // The role's own arrays: static LDS in the ordinary instantiation (compile-time addresses: carving them from the dynamic buffer
  // cost 2-5 % at the small launch shapes); in the low-register one they live in the launch's DYNAMIC LDS, in front of the staged
  // pairs and in rows L long (100 L bytes; the torsion role's workgroups use the same bytes for their history), so that the
  // kernel's static LDS is 9 KB instead of 35 and two workgroups share a CU up to 256 residues.
  // (... and in the 512-thread instantiation, whose torsion role stages its history in up to 128 KB of dynamic LDS: ARR_DYN)
  constexpr bool ARR_DYN = LOWREG || NT > 256;
  __shared__ float st_xyz[ARR_DYN ? 4 : NT * 16];
  __shared__ float st_dt[ARR_DYN ? 4 : NT * 6];
  __shared__ float st_gp[ARR_DYN ? 4 : NT * 3];
  lds_f* const s_xyz = ARR_DYN ? (lds_f*)s_hist : (lds_f*)st_xyz;   // [.][16] trial coordinates (neighbours read each other's)
  lds_f* const s_dt = ARR_DYN ? s_xyz + (size_t)L * 16 : (lds_f*)st_dt;  // [.][6] gradient a residue's psi, omega and link terms put on N and CA of the residue after it
  lds_f* const s_gp = ARR_DYN ? s_dt + (size_t)L * 6 : (lds_f*)st_gp;    // [.][3] gradient a residue's phi and backbone H put on C of the residue before it
  lds_f4* const c_hist = (lds_f4*)s_hist + (ARR_DYN ? CART_ARRAYS_BYTES(L) / 16 : 0);
  int* gi = A.st_i + (size_t)dec * SI_N;
  double* gd_ = A.st_d + (size_t)dec * SD_N;
  KSTAMP_DECL
  const int seq = *A.seq_ctr;
  for (int i = tid; i < A.nruns * (int)(sizeof(trx2_run) / 4); i += NT) s_runs[i] = reinterpret_cast<const int*>(A.runs)[i];
  if (tid < SI_N && tid >= 2) s_i[tid] = gi[tid];
  if (tid == 0) {  // (run, seq) in ONE 8-byte load
    const unsigned long long rs = *reinterpret_cast<const volatile unsigned long long*>(gi);
    s_i[SI_RUN] = (int)(unsigned)(rs & 0xffffffffull); s_i[SI_SEQ] = (int)(unsigned)(rs >> 32);
  }
  if (tid < SD_N) s_d[tid] = gd_[tid];
  if (tid < LBM) s_rho[tid] = A.rho[(size_t)dec * LBM + tid];
  const int nsl_pre = A.nslice[min(tid, L - 1)];       // slices of this residue's row; backbone H of this residue: requested with the state
  const bool hH_me = A.hasH[min(tid, L - 1)] != 0;
  float4 e_rp[3];
  { const float4* rpp = rama_par_ptr(A.hasH, L) + (size_t)min(tid, L - 1) * 3; e_rp[0] = rpp[0]; e_rp[1] = rpp[1]; e_rp[2] = rpp[2]; }
  float4 e_fa[1][6];   // the first record slice of this residue, requested with the state (sum_pair_records)
  load_record(A.FA, 0, A.B, L, dec, min(tid, L - 1), e_fa[0]);
  // The decoy's Gram scalars (kernel_step.h, "Gram form"): one set per decoy serves both roles -- a decoy is in one role at a
  // time and every run starts from an empty history, whose scalars are zeros.
  double gr_old0 = 0, gr_old1 = 0;
  // Chains of more than 256 residues (512 threads, 256 registers per thread, no staged history) keep the two-loop form: the
  // Gram form holds the new pair and two buffered pairs beside the recurrences and spilled 204 registers there.
  constexpr bool CG = NT <= 256;
  if (CG) {
    if (tid < 2 * LBM * LBM) { gr_old0 = A.gram[(size_t)dec * GR_N + tid]; s_gl.gram[tid] = gr_old0; }
    if (tid < 2 * LBM) { gr_old1 = A.gram[(size_t)dec * GR_N + 2 * LBM * LBM + tid]; s_gl.gram[2 * LBM * LBM + tid] = gr_old1; }
  }
  bool gram_dirty = false;
  __syncthreads();
  const trx2_run* runs_l = reinterpret_cast<const trx2_run*>(s_runs);
  int run = s_i[SI_RUN], phase = s_i[SI_PHASE];
  if (phase == PH_DONE || phase == PH_REPORT || s_i[SI_SEQ] == seq) return;  // the report is the torsion role's
  const trx2_run R = runs_l[min(run, A.nruns - 1)];
  if (!R.cartesian) return;
  const size_t vb = (size_t)dec * L;
  // The stored pairs, newest first, requested into LDS now and read by the recursion a phase later (as the torsion role does):
  // slot t = the t-th newest pair, [s | y][4 components][L] float4.  The rows are exactly L long -- lanes beyond the chain
  // are switched off, LDS-DMA writes only for active lanes -- so that 7 pairs of a 150-residue chain fit beside the
  // kernel's static LDS; what does not fit is read from global memory in the recursion.
  const int hl0 = s_i[SI_HL], hh0 = s_i[SI_HH];
  const int nl = (NT <= 256) ? min(hl0, A.hist_lds) : 0;
  if (NT <= 256 && act) {
    const int wave = tid >> 6;
    for (int kk = 0; kk < nl; kk++) {
      const int j = (hh0 - 1 - kk + LBM) % LBM;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        lds_dma16(A.CS + (((size_t)dec * LBM + j) * 4 + q) * L + r, (float4*)(c_hist + (size_t)((kk * 2 + 0) * 4 + q) * L + wave * 64));
        lds_dma16(A.CY + (((size_t)dec * LBM + j) * 4 + q) * L + r, (float4*)(c_hist + (size_t)((kk * 2 + 1) * 4 + q) * L + wave * 64));
      }
    }
  }
  CSTAMP(16)  // state load, role test
  CCOUNT(28)

  // ---- trial coordinates; neighbours through LDS
  float4 xt[4], gt[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { xt[q] = gt[q] = make_float4(0, 0, 0, 0); }
  double esum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  f3 gH = mk3(0, 0, 0);
  if (act) {
    const float4* xp = A.P + (vb + r) * 5;
#pragma unroll
    for (int q = 0; q < 4; q++) { xt[q] = xp[q]; lds_put((lds_f4*)(s_xyz + r * 16) + q, xt[q]); }
    float g[PR_NCOMP], ep[6];
    sum_pair_records<1>(A.FA, nsl_pre, A.B, L, dec, r, g, ep, e_fa);
    esum[0] += ep[0]; esum[1] += ep[1]; esum[2] += ep[2]; esum[3] += ep[3]; esum[4] += ep[4]; esum[8] += ep[5];
    gt[0] = make_float4(g[0], g[1], g[2], g[3]); gt[1] = make_float4(g[4], g[5], g[6], g[7]);
    gt[2] = make_float4(g[8], g[9], g[10], g[11]); gt[3] = make_float4(g[12], g[13], g[14], 0.0f);
    gH = mk3(g[15], g[16], g[17]);
  }
  __syncthreads();
  CSTAMP(17)  // coordinates -> LDS, slab sums
  f3 aN = mk3(0, 0, 0), aCA = aN, aC = aN, aO = aN, aCB = aN;  // gradient of the local terms on this residue's atoms
  Res5 Me = unpack5(s_xyz + (act ? r : 0) * 16), Pv = Me, Nx = Me;
  if (act && r > 0) Pv = unpack5(s_xyz + (r - 1) * 16);
  if (act && r + 1 < L) Nx = unpack5(s_xyz + (r + 1) * 16);
  // Every term that spans the peptide bond is evaluated ONCE, by the residue on its N-terminal side (phi: by its own residue),
  // and the part of its gradient that lands on the neighbour's atoms is handed over through LDS: three floats to C of the
  // residue before (phi; the backbone H), six to N and CA of the residue after (psi, omega, the link terms).  Both owners
  // evaluated such a term before -- up to six more dihedral gradients and a second set of link terms per thread.
  if (act) {
    f3 toC = mk3(0, 0, 0), toN = toC, toCA = toC;
    // The pair kernel's gradient on the backbone H goes to the three atoms H is built from (transpose of the Jacobian of
    // place_h; oracle: hb_spread_h): N and CA of this residue, C of the previous one.
    if (R.w[7] != 0.0f && r > 0) {
      const f3 d1 = Me.N - Pv.C, d2 = Me.N - Me.CA;
      const float il1 = rsqrtf(dot(d1, d1)), il2 = rsqrtf(dot(d2, d2));
      const f3 e1 = d1 * il1, e2 = d2 * il2, u = e1 + e2;
      const float ilu = rsqrtf(dot(u, u));
      const f3 uh = u * ilu;
      const f3 gp = (gH - uh * dot(uh, gH)) * ((float)TRX2_HB_B_NH * ilu);
      const f3 g1 = (gp - e1 * dot(e1, gp)) * il1, g2 = (gp - e2 * dot(e2, gp)) * il2;
      aN = gH + g1 + g2; aCA = g2 * -1.0f; toC = g1 * -1.0f;
    }
    // rama (residues 2..L-1) and omega_bb (peptides 1..L-1; the fitted tether's centre and width follow psi)
    f3 t1, t2, t3, t4;
    if (r < L - 1) {
      const float ps = dihedral_grad(Me.N, Me.CA, Me.C, Nx.N, t1, t2, t3, t4);
      float sps, cps, dpsi = 0.0f;
      fast_sincosf(ps, &sps, &cps);
      if (r >= 1) {
        f3 p1, p2, p3, p4;
        const float ph = dihedral_grad(Pv.C, Me.N, Me.CA, Me.C, p1, p2, p3, p4);
        float sph, cph, gph, gps;
        fast_sincosf(ph, &sph, &cph);
        esum[5] += (double)rama_eval(sph, cph, sps, cps, e_rp, gph, gps);
        const float dphi = R.w[4] * gph;
        dpsi = R.w[4] * gps;
        toC = fma3(p1, dphi, toC); aN = fma3(p2, dphi, aN); aCA = fma3(p3, dphi, aCA); aC = fma3(p4, dphi, aC);
      }
      // the omega term's part on psi joins the rama term's before psi's gradient vectors go (value first: one set of vectors live at a time)
      float gps_o, gom;
      esum[6] += (double)omega_eval(sps, cps, dihedral_val(Me.CA, Me.C, Nx.N, Nx.CA), gps_o, gom);
      dpsi += R.w[5] * gps_o;
      const float dom = R.w[5] * gom;
      aN = fma3(t1, dpsi, aN); aCA = fma3(t2, dpsi, aCA); aC = fma3(t3, dpsi, aC); toN = fma3(t4, dpsi, toN);
      dihedral_grad(Me.CA, Me.C, Nx.N, Nx.CA, t1, t2, t3, t4);
      aCA = fma3(t1, dom, aCA); aC = fma3(t2, dom, aC); toN = fma3(t3, dom, toN); toCA = fma3(t4, dom, toCA);
    }
    CSTAMP(18)  // backbone H, rama / omega
    // bonded term: ideal CB geometry from the ideal local frame
    const float wcb = R.w[6];
    if (wcb != 0.0f) {
      const float d_cacb = c_cb_ideal[0], a_ncacb = c_cb_ideal[1], a_ccacb = c_cb_ideal[2], t_cb = c_cb_ideal[3];
      const float KL = (float)TRX2_CART_KLEN, KA = (float)TRX2_CART_KANG, KI = (float)TRX2_CART_KIMP;
      f3 bN = mk3(0, 0, 0), bCA = bN, bC = bN, bO = bN, bCB = bN, a, b, c, d;
      float eb = hbond(Me.N, Me.CA, (float)TRX2_B_N_CA, KL, a); bN += a; bCA += a * -1.0f;
      eb += hbond(Me.CA, Me.C, (float)TRX2_B_CA_C, KL, a); bCA += a; bC += a * -1.0f;
      eb += hbond(Me.C, Me.O, (float)TRX2_B_C_O, KL, a); bC += a; bO += a * -1.0f;
      eb += hbond(Me.CA, Me.CB, d_cacb, KL, a); bCA += a; bCB += a * -1.0f;
      eb += hangle(Me.N, Me.CA, Me.C, (float)TRX2_A_N_CA_C * TRX2_DEG_F, KA, a, b, c); bN += a; bCA += b; bC += c;
      eb += hangle(Me.CA, Me.C, Me.O, (float)TRX2_A_CA_C_O * TRX2_DEG_F, KA, a, b, c); bCA += a; bC += b; bO += c;
      eb += hangle(Me.N, Me.CA, Me.CB, a_ncacb, KA, a, b, c); bN += a; bCA += b; bCB += c;
      eb += hangle(Me.C, Me.CA, Me.CB, a_ccacb, KA, a, b, c); bC += a; bCA += b; bCB += c;
      eb += hdih(Me.N, Me.C, Me.CA, Me.CB, t_cb, KI, a, b, c, d); bN += a; bC += b; bCA += c; bCB += d;
      if (r + 1 < L) {
        const LinkGrad G = link_terms(Me, Nx);
        eb += G.e; bCA += G.CA; bC += G.C; bO += G.O;
        toN = fma3(G.Nn, wcb, toN); toCA = fma3(G.CAn, wcb, toCA);
      }
      esum[7] += (double)eb;
      aN = fma3(bN, wcb, aN); aCA = fma3(bCA, wcb, aCA); aC = fma3(bC, wcb, aC); aO = fma3(bO, wcb, aO); aCB = fma3(bCB, wcb, aCB);
    }
    s_gp[r * 3] = toC.x; s_gp[r * 3 + 1] = toC.y; s_gp[r * 3 + 2] = toC.z;
    s_dt[r * 6] = toN.x; s_dt[r * 6 + 1] = toN.y; s_dt[r * 6 + 2] = toN.z;
    s_dt[r * 6 + 3] = toCA.x; s_dt[r * 6 + 4] = toCA.y; s_dt[r * 6 + 5] = toCA.z;
  }
  __syncthreads();
  CSTAMP(19)  // bonded term, hand-over
  if (act) {
    if (r + 1 < L) aC += mk3(s_gp[(r + 1) * 3], s_gp[(r + 1) * 3 + 1], s_gp[(r + 1) * 3 + 2]);
    if (r > 0) {
      aN += mk3(s_dt[(r - 1) * 6], s_dt[(r - 1) * 6 + 1], s_dt[(r - 1) * 6 + 2]);
      aCA += mk3(s_dt[(r - 1) * 6 + 3], s_dt[(r - 1) * 6 + 4], s_dt[(r - 1) * 6 + 5]);
    }
    gt[0].x += aN.x; gt[0].y += aN.y; gt[0].z += aN.z; gt[0].w += aCA.x;
    gt[1].x += aCA.y; gt[1].y += aCA.z; gt[1].z += aC.x; gt[1].w += aC.y;
    gt[2].x += aC.z; gt[2].y += aO.x; gt[2].z += aO.y; gt[2].w += aO.z;
    gt[3].x += aCB.x; gt[3].y += aCB.y; gt[3].z += aCB.z;
  }
  CSTAMP(20)  // neighbours' parts, gradient assembled
  // the minimiser needs the weighted total only (the terms are reported by the torsion role's report evaluation): ONE f64
  // workgroup sum instead of nine
  double ft1[1] = {(double)R.w[0] * esum[0] + (double)R.w[1] * (esum[1] + esum[2]) + (double)R.w[2] * esum[3] + (double)R.w[3] * esum[4] +
                   (double)R.w[4] * esum[5] + (double)R.w[5] * esum[6] + (double)R.w[6] * esum[7] + (double)R.w[7] * esum[8]};
  block_sum_n<1, NW>(ft1, s_buf, flip);
  const double f_t = uniform_d(ft1[0]);
  if (tid == 0) A.f_last[dec] = f_t;
#ifdef TRX2_SELFCHECK
  {  // checking build only: the Cartesian role's one sum against the nine terms reduced one by one (g_selfcheck[2], [3])
    double chk[9];
#pragma unroll
    for (int k = 0; k < 9; k++) chk[k] = esum[k];
    block_sum_n<9, NW>(chk, s_buf, flip);
    const double f9 = (double)R.w[0] * chk[0] + (double)R.w[1] * (chk[1] + chk[2]) + (double)R.w[2] * chk[3] + (double)R.w[3] * chk[4] +
                      (double)R.w[4] * chk[5] + (double)R.w[5] * chk[6] + (double)R.w[6] * chk[7] + (double)R.w[7] * chk[8];
    if (tid == 0) {
      atomicAdd(&g_selfcheck[2], 1ull);
      if (!(fabs(f9 - f_t) <= 1e-9 * fabs(f9) + 1e-9)) atomicAdd(&g_selfcheck[3], 1ull);
    }
  }
#endif

  // ------------------------------------------------------------------ minimiser state machine (as k_chain, 4 float4 per residue)
  int iter = s_i[SI_ITER], nls = s_i[SI_NLS], hl = s_i[SI_HL], hh = s_i[SI_HH], nh = s_i[SI_NH];
  int n_evals = s_i[SI_NEVALS] + 1, n_iters = s_i[SI_NITERS], status = s_i[SI_STATUS];
  double f = s_d[SD_F], alpha = s_d[SD_ALPHA], gdir = s_d[SD_GD];
  double fh[3] = {s_d[SD_FH0], s_d[SD_FH1], s_d[SD_FH2]};
  double gamma_h = s_d[SD_GAMMA];
  int wspace = s_i[SI_WSPACE];
  bool started = false;
  float4 x[4], g[4], dv[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    x[q] = g[q] = dv[q] = make_float4(0, 0, 0, 0);
    if (act) { x[q] = A.CX[(vb + r) * 4 + q]; g[q] = A.CG[(vb + r) * 4 + q]; dv[q] = A.CD[(vb + r) * 4 + q]; }
  }
  bool next_run = false, new_dir = false, steepest = false, new_trial = false;
  float4 sv[4], yv[4];  // the pair of the step just accepted: the newest pair of the recursion, taken from registers
  bool stored = false;
#pragma unroll
  for (int q = 0; q < 4; q++) sv[q] = yv[q] = make_float4(0, 0, 0, 0);
  const bool finite_t = isfinite(f_t);
  CSTAMP(21)  // energy reduction, loads of X, G, D
  if (!finite_t && phase == PH_START) { status = TRX2_DIVERGED; phase = PH_DONE; }
  else if (phase == PH_START) {
    f = f_t;
#pragma unroll
    for (int q = 0; q < 4; q++) { x[q] = xt[q]; g[q] = gt[q]; }
    hl = 0; hh = 0; nh = 1; fh[0] = f; iter = 0;
    steepest = true; started = true;
  } else {
    double fref = fh[0];
    for (int k = 1; k < nh; k++) fref = fmax(fref, fh[k]);
    const bool accept = finite_t && f_t <= fref + (double)TRX2_LS_C1 * alpha * gdir;
    if (accept) {
      double v3[3] = {0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        sv[q] = make_float4(xt[q].x - x[q].x, xt[q].y - x[q].y, xt[q].z - x[q].z, xt[q].w - x[q].w);
        yv[q] = make_float4(gt[q].x - g[q].x, gt[q].y - g[q].y, gt[q].z - g[q].z, gt[q].w - g[q].w);
      }
      if constexpr (CG) {
        // Every product of the new pair and the new gradient with the stored pairs, ONE fused reduction (as the torsion role;
        // the two-loop form made 2 x 8 dependent rounds of { dot, workgroup sum, axpy } here: 42 % of this role's step at
        // L = 150, profiles/README.md round 3).  The stored pairs by their age BEFORE this step: age k < nl was staged in
        // LDS slot k at the top of the step, the others come from global memory, two pairs at a time.  Everything is indexed
        // by compile-time constants (a runtime index would move pv[] to scratch); the tests on hl / nl are wave-uniform.
        if (NT <= 256) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the staged pairs have landed (long ago)
        float pv[GV_N];
#pragma unroll
        for (int q = 0; q < GV_N; q++) pv[q] = 0.0f;
        const int rc = min(r, L - 1);
        if constexpr (LOWREG) {
#pragma unroll
          for (int k = 0; k < LBM; k++) {
            if (k < hl) {
              float4 cs_[4], cy_[4];
              cart_hist_fetch<NT>(A, c_hist, dec, L, rc, k < nl, k, (hh - 1 - k + 2 * LBM) % LBM, cs_, cy_);
              float a = 0, b = 0, c = 0;
#pragma unroll
              for (int q = 0; q < 4; q++) { a += dot4(cs_[q], yv[q]); b += dot4(cy_[q], yv[q]); c += dot4(sv[q], cy_[q]); }
              pv[GV_A + k] = act ? a : 0.0f; pv[GV_B + k] = act ? b : 0.0f;
              if (k < LBM - 1) pv[GV_C + k] = act ? c : 0.0f;
            }
            __builtin_amdgcn_sched_barrier(0);  // one pair's registers at a time
          }
        } else {
        // two named buffers: pair k + 1 is requested before pair k is consumed
          float4 b0s[4], b0y[4], b1s[4], b1y[4];
          if (0 < hl) cart_hist_fetch<NT>(A, c_hist, dec, L, rc, 0 < nl, 0, (hh - 1 + LBM) % LBM, b0s, b0y);
#pragma unroll
          for (int k = 0; k < LBM; k++) {
            float4 (&cs_)[4] = (k & 1) ? b1s : b0s; float4 (&cy_)[4] = (k & 1) ? b1y : b0y;
            float4 (&ns_)[4] = (k & 1) ? b0s : b1s; float4 (&ny_)[4] = (k & 1) ? b0y : b1y;
            if (k + 1 < LBM && k + 1 < hl) cart_hist_fetch<NT>(A, c_hist, dec, L, rc, k + 1 < nl, k + 1, (hh - 2 - k + 2 * LBM) % LBM, ns_, ny_);
            if (k < hl) {
              float a = 0, b = 0, c = 0;
#pragma unroll
              for (int q = 0; q < 4; q++) { a += dot4(cs_[q], yv[q]); b += dot4(cy_[q], yv[q]); c += dot4(sv[q], cy_[q]); }
              // idle threads hold copies of the last residue: selected out
              pv[GV_A + k] = act ? a : 0.0f; pv[GV_B + k] = act ? b : 0.0f;
              if (k < LBM - 1) pv[GV_C + k] = act ? c : 0.0f;
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {  // (s, y, gt are zero in idle threads)
          pv[GV_SY] += dot4(sv[q], yv[q]); pv[GV_SS] += dot4(sv[q], sv[q]); pv[GV_YY] += dot4(yv[q], yv[q]);
          pv[GV_SG] += dot4(sv[q], gt[q]); pv[GV_YG] += dot4(yv[q], gt[q]); pv[GV_GG] += dot4(gt[q], gt[q]);
        }
        gram_reduce<NT>(pv, s_gl);
        v3[0] = s_gl.out[GV_SY]; v3[1] = s_gl.out[GV_SS]; v3[2] = s_gl.out[GV_YY];
        if (v3[0] > 1e-12 * sqrt(v3[1] * v3[2])) {
          if (act)
#pragma unroll
            for (int q = 0; q < 4; q++) {
              A.CS[(((size_t)dec * LBM + hh) * 4 + q) * L + r] = sv[q];
              A.CY[(((size_t)dec * LBM + hh) * 4 + q) * L + r] = yv[q];
            }
          stored = true;
          gamma_h = v3[0] / v3[2];
          wspace = 2;
          hh = (hh + 1) % LBM;
          if (hl < LBM) hl++;
        }
        gram_advance<NT>(s_gl, stored, gr_old0, gr_old1);  // ends with a barrier
        gram_dirty = true;
      } else {
#pragma unroll
      for (int q = 0; q < 4; q++) { v3[0] += (double)dot4(sv[q], yv[q]); v3[1] += (double)dot4(sv[q], sv[q]); v3[2] += (double)dot4(yv[q], yv[q]); }
      block_sum_n<3, NW>(v3, s_buf, flip);
      if (v3[0] > 1e-12 * sqrt(v3[1] * v3[2])) {
        if (act)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            A.CS[(((size_t)dec * LBM + hh) * 4 + q) * L + r] = sv[q];
            A.CY[(((size_t)dec * LBM + hh) * 4 + q) * L + r] = yv[q];
          }
        stored = true;
        __syncthreads();
        if (tid == 0) s_rho[hh] = (float)(1.0 / v3[0]);
        gamma_h = v3[0] / v3[2];
        wspace = 2;
        __syncthreads();
        hh = (hh + 1) % LBM;
        if (hl < LBM) hl++;
      }
      }
      const double fprev = f;
#pragma unroll
      for (int q = 0; q < 4; q++) { x[q] = xt[q]; g[q] = gt[q]; }
      f = f_t;
      if (nh < TRX2_LS_PAST) fh[nh++] = f;
      else { fh[0] = fh[1]; fh[1] = fh[2]; fh[2] = f; }
      iter++; n_iters++;
      const bool conv = 2.0 * fabs(fprev - f) <= (R.tol > 0.0f ? (double)R.tol : (double)TRX2_MIN_TOL) * (fabs(fprev) + fabs(f) + 1e-10);
      if (conv || iter >= R.max_iter) next_run = true;
      else new_dir = true;
    } else {
      nls++;
      alpha *= (double)TRX2_LS_SHRINK;
      if (nls > TRX2_LS_MAXTRIAL) {
        if (hl > 0) { hl = 0; steepest = true; }
        else next_run = true;
      } else new_trial = true;
    }
  }
  CSTAMP(22)  // Armijo / (s, y) pair
  if (new_dir) {
    CCOUNT(29)
    if constexpr (CG) {
    // The recurrences on the Gram scalars (every thread, no barrier), then ONE pass over the stored vectors: r = gamma g + sum_m
    // cy_m y_m + cs_m s_m with m the age AFTER this step -- m = 0 is the pair stored a moment ago (registers), age m was age
    // m - 1 before (staged slot m - 1, or global memory).
    float4 qv[4];
    {
      float cy[LBM], cs[LBM];
      double g_r;
      gram_recursion(s_gl.gram, hl, gamma_h, s_gl.out[GV_GG], cy, cs, g_r);
      CSTAMP(23)  // recurrences on the Gram scalars
      const float gam = (float)gamma_h;
#pragma unroll
      for (int q = 0; q < 4; q++) qv[q] = make_float4(gam * g[q].x, gam * g[q].y, gam * g[q].z, gam * g[q].w);
      const int rc = min(r, L - 1), sh = stored ? 1 : 0;
      if constexpr (LOWREG) {
#pragma unroll
        for (int m = 0; m < LBM; m++) {
          if (m < hl) {
            float4 cs_[4], cy_[4];
            if (m == 0 && sh) {
#pragma unroll
              for (int q = 0; q < 4; q++) { cs_[q] = sv[q]; cy_[q] = yv[q]; }
            } else cart_hist_fetch<NT>(A, c_hist, dec, L, rc, m - sh < nl, m - sh, (hh - 1 - m + 2 * LBM) % LBM, cs_, cy_);
            const float a = cs[m], b = cy[m];
#pragma unroll
            for (int q = 0; q < 4; q++) {
              qv[q].x = fmaf(b, cy_[q].x, fmaf(a, cs_[q].x, qv[q].x)); qv[q].y = fmaf(b, cy_[q].y, fmaf(a, cs_[q].y, qv[q].y));
              qv[q].z = fmaf(b, cy_[q].z, fmaf(a, cs_[q].z, qv[q].z)); qv[q].w = fmaf(b, cy_[q].w, fmaf(a, cs_[q].w, qv[q].w));
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
      float4 b0s[4], b0y[4], b1s[4], b1y[4];
        if (sh) {
#pragma unroll
          for (int q = 0; q < 4; q++) { b0s[q] = sv[q]; b0y[q] = yv[q]; }
        } else if (0 < hl) cart_hist_fetch<NT>(A, c_hist, dec, L, rc, 0 < nl, 0, (hh - 1 + LBM) % LBM, b0s, b0y);
#pragma unroll
        for (int m = 0; m < LBM; m++) {
          float4 (&cs_)[4] = (m & 1) ? b1s : b0s; float4 (&cy_)[4] = (m & 1) ? b1y : b0y;
          float4 (&ns_)[4] = (m & 1) ? b0s : b1s; float4 (&ny_)[4] = (m & 1) ? b0y : b1y;
          if (m + 1 < LBM && m + 1 < hl) cart_hist_fetch<NT>(A, c_hist, dec, L, rc, m + 1 - sh < nl, m + 1 - sh, (hh - 2 - m + 2 * LBM) % LBM, ns_, ny_);
          if (m < hl) {
            const float a = cs[m], b = cy[m];
#pragma unroll
            for (int q = 0; q < 4; q++) {
              qv[q].x = fmaf(b, cy_[q].x, fmaf(a, cs_[q].x, qv[q].x)); qv[q].y = fmaf(b, cy_[q].y, fmaf(a, cs_[q].y, qv[q].y));
              qv[q].z = fmaf(b, cy_[q].z, fmaf(a, cs_[q].z, qv[q].z)); qv[q].w = fmaf(b, cy_[q].w, fmaf(a, cs_[q].w, qv[q].w));
            }
          }
        }
      }
      CSTAMP(25)  // direction from the stored vectors
      const float4 z4 = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; q++) dv[q] = act ? make_float4(-qv[q].x, -qv[q].y, -qv[q].z, -qv[q].w) : z4;
      const double gd = -g_r, gg = s_gl.out[GV_GG];
      if (!(gg > 0)) next_run = true;
      else if (hl == 0 || !(gd < 0)) { hl = 0; steepest = true; }
      else { gdir = gd; alpha = 1.0; nls = 0; new_trial = true; }
    }
    } else {
    // Chains of 257-512 residues: two-loop recursion on global memory, 8 float4 per thread and stored pair.  Loads are branch-free
    // (index clamped to the last residue; idle threads are masked out of the dot and the update instead) so that a pair's eight
    // loads issue together -- guarded per element, each load had its own branch and wait (~2000 cycles per round: s_memtime
    // stamps, profiles/README.md).
    float4 qv[4];
#pragma unroll
    for (int q = 0; q < 4; q++) qv[q] = g[q];
    const int rc = min(r, L - 1);
    auto pair_at = [&](int kk) { return (hh - 1 - kk + LBM) % LBM; };
    auto load_pair = [&](int kk, float4 (&s_)[4], float4 (&y_)[4]) {
      const size_t o = ((size_t)dec * LBM + pair_at(kk)) * 4 * L + rc;
#pragma unroll
      for (int q = 0; q < 4; q++) { s_[q] = A.CS[o + (size_t)q * L]; y_[q] = A.CY[o + (size_t)q * L]; }
    };
    auto round1 = [&](int kk, const float4 (&s_)[4], const float4 (&y_)[4]) {
      const int j = pair_at(kk);
      double v1[1] = {0};
#pragma unroll
      for (int q = 0; q < 4; q++) v1[0] += (double)dot4(s_[q], qv[q]);  // qv is zero in idle threads
      block_sum_n<1, NW>(v1, s_buf, flip);
      const float al = s_rho[j] * (float)v1[0];
      if (tid == 0) s_alpha[j] = al;
      const float am = act ? al : 0.0f;
#pragma unroll
      for (int q = 0; q < 4; q++) { qv[q].x -= am * y_[q].x; qv[q].y -= am * y_[q].y; qv[q].z -= am * y_[q].z; qv[q].w -= am * y_[q].w; }
    };
    auto round2 = [&](int kk, const float4 (&s_)[4], const float4 (&y_)[4]) {
      const int j = pair_at(kk);
      double v1[1] = {0};
#pragma unroll
      for (int q = 0; q < 4; q++) v1[0] += (double)dot4(y_[q], qv[q]);
      block_sum_n<1, NW>(v1, s_buf, flip);
      const float c = act ? s_alpha[j] - s_rho[j] * (float)v1[0] : 0.0f;
#pragma unroll
      for (int q = 0; q < 4; q++) { qv[q].x += c * s_[q].x; qv[q].y += c * s_[q].y; qv[q].z += c * s_[q].z; qv[q].w += c * s_[q].w; }
    };
    {
      // Three rotating buffers, loops advancing in groups of three so that the buffer of pair kk (kk % 3) is a compile-time
      // choice: two pairs are always in flight.  With 64 decoys the histories (460 KB each at L=150) do not stay in L2 and a
      // round is shorter than one memory round trip, so one pair ahead was not enough (profiles/README.md).
      float4 sb[3][4], yb[3][4];
      if (hl > 0) load_pair(0, sb[0], yb[0]);
      if (hl > 1) load_pair(1, sb[1], yb[1]);
#pragma unroll 1
      for (int k0 = 0; k0 < hl; k0 += 3)
#pragma unroll
        for (int u = 0; u < 3; u++) {
          const int kk = k0 + u;
          if (kk < hl) {
            if (kk + 2 < hl) load_pair(kk + 2, sb[(u + 2) % 3], yb[(u + 2) % 3]);
            round1(kk, sb[u], yb[u]);
          }
        }
      CSTAMP(23)  // two-loop: first loop
      if (hl > 0) {
        const float gam = (float)gamma_h;
#pragma unroll
        for (int q = 0; q < 4; q++) { qv[q].x *= gam; qv[q].y *= gam; qv[q].z *= gam; qv[q].w *= gam; }
      }
      __syncthreads();
      CSTAMP(24)  // two-loop: gamma
      // backwards: the three oldest pairs are still in their buffers; a buffer is refilled (pair kk - 3) as soon as it is done
#pragma unroll 1
      for (int k0 = (hl - 1) / 3 * 3; k0 >= 0; k0 -= 3)
#pragma unroll
        for (int u = 2; u >= 0; u--) {
          const int kk = k0 + u;
          if (kk < hl) {
            round2(kk, sb[u], yb[u]);
            if (kk >= 3) load_pair(kk - 3, sb[u], yb[u]);
          }
        }
    }
    CSTAMP(25)  // two-loop: second loop
    double v2[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      dv[q] = make_float4(-qv[q].x, -qv[q].y, -qv[q].z, -qv[q].w);
      v2[0] += (double)dot4(g[q], dv[q]); v2[1] += (double)dot4(g[q], g[q]);
    }
    block_sum_n<2, NW>(v2, s_buf, flip);
    if (!(v2[1] > 0)) next_run = true;
    else if (hl == 0 || !(v2[0] < 0)) { hl = 0; steepest = true; }
    else { gdir = v2[0]; alpha = 1.0; nls = 0; new_trial = true; }
    }
  }
  if (steepest) {
    double v1[1] = {0};
#pragma unroll
    for (int q = 0; q < 4; q++) { dv[q] = make_float4(-g[q].x, -g[q].y, -g[q].z, -g[q].w); v1[0] += (double)dot4(g[q], g[q]); }
    block_sum_n<1, NW>(v1, s_buf, flip);
    if (!(v1[0] > 0)) next_run = true;
    else {
      gdir = -v1[0]; nls = 0; new_trial = true;
      alpha = ((R.precheck & TRX2_RUN_WARM) && started && wspace == 2 && gamma_h > 0.0) ? fmin(gamma_h, 1.0 / sqrt(v1[0])) : fmin(1.0, 1.0 / sqrt(v1[0]));
    }
  }
  if (next_run) {
    run++;
    phase = (run >= A.nruns) ? PH_DONE : PH_START;
#pragma unroll
    for (int q = 0; q < 4; q++) xt[q] = x[q];  // coordinates of the accepted point go back into the xyz buffer
  }
  if (new_trial) {
    phase = PH_LS;
    const float al = (float)alpha;
#pragma unroll
    for (int q = 0; q < 4; q++)
      xt[q] = make_float4(fmaf(al, dv[q].x, x[q].x), fmaf(al, dv[q].y, x[q].y), fmaf(al, dv[q].z, x[q].z), fmaf(al, dv[q].w, x[q].w));
  }
  if (phase != PH_DONE && n_evals >= A.max_evals) { status = TRX2_MAXEVAL; phase = PH_DONE; }
  const bool over = phase == PH_DONE;  // protocol finished here, budget spent, or diverged: the report (torsion role) comes next
  if (over) {
    phase = PH_REPORT; run = A.nruns - 1;
#pragma unroll
    for (int q = 0; q < 4; q++) xt[q] = x[q];  // at the accepted point
  }
  // ---- store state and the coordinates for the next pair launch
  if (act) {
#pragma unroll
    for (int q = 0; q < 4; q++) { A.CX[(vb + r) * 4 + q] = x[q]; A.CG[(vb + r) * 4 + q] = g[q]; A.CD[(vb + r) * 4 + q] = dv[q]; }
    float4* xo = A.P + (vb + r) * 5;
#pragma unroll
    for (int q = 0; q < 4; q++) xo[q] = xt[q];
    if (A.xyzT) {
      float4* xT = A.xyzT + ((size_t)((dec / A.BW) * L + r) * 5) * A.BW + dec % A.BW;
      float4 t0, t1, t2, t3;
      xt_pack_from_p(xt[0], xt[1], xt[2], xt[3], t0, t1, t2, t3);
      xT[0] = t0; xT[A.BW] = t1; xT[2 * A.BW] = t2; xT[3 * A.BW] = t3;
    }
  }
  {  // backbone H of the new trial point: C of the previous residue comes through LDS
    __syncthreads();
    if (act)
#pragma unroll
      for (int q = 0; q < 4; q++) lds_put((lds_f4*)(s_xyz + r * 16) + q, xt[q]);
    __syncthreads();
    if (act) {
      const Res5 M3 = unpack5(s_xyz + r * 16);
      float4 hv = make_float4(M3.N.x + 1.0f, M3.N.y, M3.N.z, 0.0f);
      if (r > 0) {
        const Res5 P3 = unpack5(s_xyz + (r - 1) * 16);
        const f3 Hn = place_h(P3.C, M3.N, M3.CA);
        hv = make_float4(Hn.x, Hn.y, Hn.z, hH_me ? 1.0f : 0.0f);
      }
      A.P[(vb + r) * 5 + 4] = hv;
      if (A.xyzT) A.xyzT[((size_t)((dec / A.BW) * L + r) * 5 + 4) * A.BW + dec % A.BW] = hv;
    }
  }
  CSTAMP(26)  // direction test, trial point, state + coordinate stores
  // ---- leaving Cartesian space (run finished, or the decoy stops here on its evaluation budget / divergence): torsions +
  //      relaxed internal geometry of the ACCEPTED point, for the torsion-space runs after it and for the final report
  if (next_run || over) {
    __syncthreads();
    if (act)
#pragma unroll
      for (int q = 0; q < 4; q++) lds_put((lds_f4*)(s_xyz + r * 16) + q, x[q]);
    __syncthreads();
    if (act) {
      const Res5 M2 = unpack5(s_xyz + r * 16);
      f3 d1, d2, d3, d4;
      ResGeom G = ideal_geom();
      G.g0.x = sqrtf(dot(M2.CA - M2.N, M2.CA - M2.N)); G.g0.y = sqrtf(dot(M2.C - M2.CA, M2.C - M2.CA));
      G.g0.w = angle_grad(M2.N, M2.CA, M2.C, d1, d2, d3);
      G.g1.z = sqrtf(dot(M2.O - M2.C, M2.O - M2.C)); G.g1.w = angle_grad(M2.CA, M2.C, M2.O, d1, d2, d3);
      {  // CB on the (b x c, b, c) basis
        f3 b = M2.CA - M2.N, c = M2.C - M2.CA, a = cross(b, c), d = M2.CB - M2.CA;
        const float bb = dot(b, b), cc = dot(c, c), bc = dot(b, c), det = bb * cc - bc * bc, db = dot(d, b), dc = dot(d, c);
        G.g2.y = dot(d, a) / dot(a, a); G.g2.z = (db * cc - dc * bc) / det; G.g2.w = (dc * bb - db * bc) / det;
      }
      const float dO = dihedral_grad(M2.N, M2.CA, M2.C, M2.O, d1, d2, d3, d4);
      float phi = TRX2_PI_F, psi = TRX2_PI_F, omg = TRX2_PI_F;
      if (r > 0) { const Res5 P2 = unpack5(s_xyz + (r - 1) * 16); phi = dihedral_grad(P2.C, M2.N, M2.CA, M2.C, d1, d2, d3, d4); }
      if (r + 1 < L) {
        const Res5 N2 = unpack5(s_xyz + (r + 1) * 16);
        G.g0.z = sqrtf(dot(N2.N - M2.C, N2.N - M2.C));
        G.g1.x = angle_grad(M2.CA, M2.C, N2.N, d1, d2, d3); G.g1.y = angle_grad(M2.C, N2.N, N2.CA, d1, d2, d3);
        psi = dihedral_grad(M2.N, M2.CA, M2.C, N2.N, d1, d2, d3, d4);
        omg = dihedral_grad(M2.CA, M2.C, N2.N, N2.CA, d1, d2, d3, d4);
      } else psi = dO - TRX2_PI_F;
      G.g2.x = wrap_pi_f(dO - psi);
      const float4 tv = make_float4(phi, psi, omg, 0);
      A.X[vb + r] = tv; A.XT[vb + r] = tv;
      A.geom[(vb + r) * 3] = G.g0; A.geom[(vb + r) * 3 + 1] = G.g1; A.geom[(vb + r) * 3 + 2] = G.g2;
    }
  }
  __syncthreads();
  if (CG && (gram_dirty || hl == 0)) {  // the scalars follow the history; a restarted history starts from zeros (they must stay finite)
    const bool z = hl == 0;
    if (tid < 2 * LBM * LBM) A.gram[(size_t)dec * GR_N + tid] = z ? 0.0 : s_gl.gram[tid];
    if (tid < 2 * LBM) A.gram[(size_t)dec * GR_N + 2 * LBM * LBM + tid] = z ? 0.0 : s_gl.gram[2 * LBM * LBM + tid];
  }
  if (tid == 0) {
    gi[SI_PHASE] = phase; gi[SI_ITER] = iter; gi[SI_NLS] = nls; gi[SI_HL] = hl; gi[SI_HH] = hh;
    gi[SI_NH] = nh; gi[SI_STATUS] = status; gi[SI_NEVALS] = n_evals; gi[SI_NITERS] = n_iters; gi[SI_WSPACE] = wspace;
    *reinterpret_cast<volatile unsigned long long*>(gi) = ((unsigned long long)(unsigned)seq << 32) | (unsigned long long)(unsigned)run;  // last, in one piece
    gd_[SD_F] = f; gd_[SD_ALPHA] = alpha; gd_[SD_GD] = gdir; gd_[SD_FH0] = fh[0]; gd_[SD_FH1] = fh[1]; gd_[SD_FH2] = fh[2];
    gd_[SD_GAMMA] = gamma_h;
    const trx2_run Rn = runs_l[min(run, A.nruns - 1)];
    float* w = A.wcur + (size_t)dec * 8;
    w[0] = Rn.w[0]; w[1] = Rn.w[1]; w[2] = Rn.w[2]; w[3] = Rn.w[3];
    w[4] = (float)Rn.sep_lo; w[5] = (float)Rn.sep_hi; w[6] = 1.0f + (float)Rn.pair_filter; w[7] = Rn.w[7];
  }
  if (tid < LBM) A.rho[(size_t)dec * LBM + tid] = s_rho[tid];
  if (NT <= 256) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA of this workgroup outlives it
}

// ---- launchable forms.  k_chain: INIT / FINISH passes and protocols without a Cartesian run.  k_step: one launch of 2B
// workgroups per evaluation -- workgroup d < B steps decoy d in Cartesian space, workgroup B + d steps it in torsion
// space; whichever does not match the decoy's current run exits at once.  The two roles touch disjoint decoys, so they
// run concurrently instead of as two half-empty launches back to back (k_cart alone was 22-27 % of GPU time).
// The torsion role runs on TN threads, RPT residues each.
template <int RPT, int TN>
__global__ __launch_bounds__(TN) void k_chain(ChainArgs A) {
  __shared__ int s_runs[STEP_RUNS_INTS];
  __shared__ GramLds<(RPT == 1) ? TN : 16> s_gl;
  chain_body<RPT, TN>(A, blockIdx.x, s_runs, s_gl);
}
template <int RPT, int TN, int NT, bool LOWREG = false>
__global__ __launch_bounds__(NT, (LOWREG ? 2 : 1)) void k_step(ChainArgs A, CartArgs C) {   // (HIP: waves per SIMD) LOWREG: two, 256 registers
  __shared__ int s_runs[STEP_RUNS_INTS];
  __shared__ GramLds<(RPT == 1) ? TN : 16> s_gl;   // either role's (a workgroup is in one)
  // the Cartesian role first: its workgroups are the slower ones, and the launch ends with the last of them
  if ((int)blockIdx.x >= A.B) {
    if (NT == TN || threadIdx.x < TN) chain_body<RPT, TN>(A, (int)blockIdx.x - A.B, s_runs, s_gl);  // the other waves of the workgroup exit at once
  } else {
    static_assert(RPT != 1 || TN == NT, "the two roles share the workgroup's Gram scratch");
    if constexpr (RPT == 1) cart_body<NT, LOWREG>(C, (int)blockIdx.x, s_runs, s_gl);
  }
}

// Shared launch (trx2fold.hip: LaunchEngine): blockIdx.y picks one of several independent folds, each with its own argument blocks
// in device memory (its map's per-residue constants, its own state, vectors, records and protocol table); blockIdx.x is the
// (role, slot) index within that fold, as in k_step.  Folds with fewer slots than the launch's widest leave at once.
template <int RPT, int TN, int NT, bool LOWREG = false>
__global__ __launch_bounds__(NT, (LOWREG ? 2 : 1)) void k_step_multi(const ChainArgs* AA, const CartArgs* CC) {
  __shared__ int s_runs[STEP_RUNS_INTS];
  __shared__ GramLds<(RPT == 1) ? TN : 16> s_gl;
  static_assert(RPT == 1 && TN == NT, "shared launches serve the one-residue-per-thread instantiations");
  const ChainArgs A = load_args(AA + blockIdx.y);
  const int nB = A.B;
  if ((int)blockIdx.x >= 2 * nB) return;
  if ((int)blockIdx.x >= nB) {
    chain_body<RPT, TN>(A, (int)blockIdx.x - nB, s_runs, s_gl);
  } else {
    const CartArgs C = load_args(CC + blockIdx.y);
    cart_body<NT, LOWREG>(C, (int)blockIdx.x, s_runs, s_gl);
  }
}
// Half-evaluation launch (launch_engine.h, round 5): ONE launch steps the single-decoy folds of group X (the second half of their
// evaluation: records -> trial point) and evaluates the pair terms of group Y (the first half of theirs) -- disjoint folds, so the
// two kinds of work have no dependency inside the launch, and the few step workgroups (one wave per SIMD of one CU per fold, a
// chain of dependent phases: latency) run beside the thousands of pair waves (throughput) instead of before or after them.  Two
// kernels on one stream cannot overlap on this device (hipExtAnyOrderLaunch is ignored on gfx9: tools/probes/anyorder_probe.hip) and a
// dependency across streams costs 11-12 us (profiles/README.md, round 4), hence one kernel with two roles:
//   workgroups [0, 2 n_step): fold = id / 2, Cartesian role (even) | torsion role (odd) -- the step kernel's low-register form
//   (256 registers: two workgroups per CU, which is what the pair role's 199 registers allow too);
//   the rest: TN / 64 independent one-wave pair work items per workgroup (pair_body<.., SUBW>), no barrier among them.
// The arithmetic of either role is that of k_step_multi<.., true> / k_pair1_multi: results are bit-identical (tests).
template <int FAM, bool SEGC, int TN>
__global__ __launch_bounds__(TN, 2) void k_half_multi(const ChainArgs* AA, const CartArgs* CC, int n_step, const PairArgs* PA, int n_pair, int max_items, int xcd_groups) {
  __shared__ int s_runs[STEP_RUNS_INTS];
  __shared__ GramLds<TN> s_gl;
  constexpr int SUBW = TN / 64;
  const unsigned nsb = 2u * (unsigned)n_step;
  if (blockIdx.x < nsb) {
    const unsigned fold = blockIdx.x >> 1;
    if (blockIdx.x & 1u) {
      const ChainArgs A = load_args(AA + fold);
      chain_body<1, TN>(A, 0, s_runs, s_gl);
    } else {
      const CartArgs C = load_args(CC + fold);
      cart_body<TN, true>(C, 0, s_runs, s_gl);
    }
    return;
  }
  const unsigned pb = blockIdx.x - nsb, sub = threadIdx.x >> 6;
  unsigned fold, item;
  if (xcd_groups) {   // all rows of a fold on workgroups with the same id modulo 8 (one XCD, one L2: k_pair1_multi)
    const unsigned x = (blockIdx.x & 7u), sidx = (pb >> 3) * SUBW + sub;
    fold = x + 8u * (sidx / (unsigned)max_items);
    item = sidx % (unsigned)max_items;
  } else {
    const unsigned id = pb * SUBW + sub;
    fold = id / (unsigned)max_items;
    item = id % (unsigned)max_items;
  }
  if ((int)fold >= n_pair) return;
  const PairArgs A = load_args(PA + fold);
  if ((int)item >= A.n_items) return;
  pair_body<1, FAM, 1, SEGC, SUBW>(A, item, 0);
}
// one pass over the folds of a shared launch: each fold's count of retired slots, gathered for ONE copy to the host
__global__ void k_gather_done(int n, const int* const* done, int* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = *done[i];
}

// ---- Tail of a fold: once the queue is empty the slots retire one by one, but a pair-kernel wave costs the same while ANY of its
// 64 decoys is alive, and a batch of several decoy groups keeps all of them partly alive almost to the end.  When no more than
// one group's worth of decoys is left, the survivors are moved into group 0 and the launches shrink to one group.
struct CompactArgs {
  int B, Bc, L, BW;          // slots before / after (Bc = BW = one group)
  int* plan;                 // [1 + 2 Bc]: count, then (from, to) pairs
  int* st_i; double* st_d; float* rho; double* gram; float* wcur; int* slot_id; int* done_count;
  float4 *X, *G, *D, *XT, *geom, *S, *Y, *P, *xyzT;
  float4 *CX, *CG, *CD, *CS, *CY;   // NULL without a Cartesian run
};
__global__ void k_compact_plan(CompactArgs A) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int nf = 0, nm = 0, live = 0;
  for (int s = 0; s < A.B; s++) live += A.st_i[(size_t)s * SI_N + SI_PHASE] != PH_DONE;
  for (int s = A.Bc; s < A.B; s++)
    if (A.st_i[(size_t)s * SI_N + SI_PHASE] != PH_DONE) {
      while (nf < A.Bc - 1 && A.st_i[(size_t)nf * SI_N + SI_PHASE] != PH_DONE) nf++;  // live <= Bc: a retired slot below Bc exists for every survivor above
      A.plan[1 + 2 * nm] = s; A.plan[2 + 2 * nm] = nf; nm++; nf++;
    }
  A.plan[0] = nm;
  *A.done_count = A.Bc - live;
}
__device__ __forceinline__ void copy_f4(float4* dst, const float4* src, size_t n) {
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_compact_move(CompactArgs A) {
  const int k = blockIdx.x;
  if (k >= A.plan[0]) return;
  const int s = A.plan[1 + 2 * k], d = A.plan[2 + 2 * k], L = A.L, tid = threadIdx.x;
  const size_t so = (size_t)s * L, dn = (size_t)d * L;
  if (tid < SI_N) A.st_i[(size_t)d * SI_N + tid] = A.st_i[(size_t)s * SI_N + tid];
  if (tid < SD_N) A.st_d[(size_t)d * SD_N + tid] = A.st_d[(size_t)s * SD_N + tid];
  if (tid < LBM) A.rho[(size_t)d * LBM + tid] = A.rho[(size_t)s * LBM + tid];
  if (tid < GR_N) A.gram[(size_t)d * GR_N + tid] = A.gram[(size_t)s * GR_N + tid];
  if (tid < 8) A.wcur[(size_t)d * 8 + tid] = A.wcur[(size_t)s * 8 + tid];
  if (tid == 0) A.slot_id[d] = A.slot_id[s];
  copy_f4(A.X + dn, A.X + so, L); copy_f4(A.G + dn, A.G + so, L); copy_f4(A.D + dn, A.D + so, L); copy_f4(A.XT + dn, A.XT + so, L);
  copy_f4(A.geom + dn * 3, A.geom + so * 3, (size_t)L * 3);
  copy_f4(A.S + dn * LBM, A.S + so * LBM, (size_t)L * LBM); copy_f4(A.Y + dn * LBM, A.Y + so * LBM, (size_t)L * LBM);
  copy_f4(A.P + dn * 5, A.P + so * 5, (size_t)L * 5);
  for (int i = tid; i < L * 5; i += blockDim.x)
    A.xyzT[((size_t)(d / A.BW) * L * 5 + i) * A.BW + d % A.BW] = A.xyzT[((size_t)(s / A.BW) * L * 5 + i) * A.BW + s % A.BW];
  if (A.CX) {
    copy_f4(A.CX + dn * 4, A.CX + so * 4, (size_t)L * 4); copy_f4(A.CG + dn * 4, A.CG + so * 4, (size_t)L * 4); copy_f4(A.CD + dn * 4, A.CD + so * 4, (size_t)L * 4);
    copy_f4(A.CS + dn * 4 * LBM, A.CS + so * 4 * LBM, (size_t)L * 4 * LBM); copy_f4(A.CY + dn * 4 * LBM, A.CY + so * 4 * LBM, (size_t)L * 4 * LBM);
  }
  __syncthreads();
  if (tid == 0) { A.st_i[(size_t)s * SI_N + SI_PHASE] = PH_DONE; A.wcur[(size_t)s * 8 + 6] = 0.0f; }  // the vacated slot (never launched again)
}

// After a compaction below one group of 64: the decoy-minor coordinate copy in the narrower layout, from the decoy-major one
// (both hold the current trial coordinates)
// (one thread per residue: the two records hold the atoms in different orders, xt_pack)
__global__ void k_relayout(int B, int L, int BW, const float4* P, float4* xyzT) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * L) return;
  const int dec = (int)(i / (size_t)L), r = (int)(i % (size_t)L);
  const float4* p = P + i * 5;
  float4 t0, t1, t2, t3;
  xt_pack_from_p(p[0], p[1], p[2], p[3], t0, t1, t2, t3);
  float4* xT = xyzT + ((size_t)((dec / BW) * L + r) * 5) * BW + dec % BW;
  xT[0] = t0; xT[BW] = t1; xT[2 * BW] = t2; xT[3 * BW] = t3; xT[4 * BW] = p[4];
}

__global__ void k_init_torsions(int L, int B, uint64_t seed, uint32_t decoy0, const float* tors0, float4* X, float4* XT, float4* geom) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L * B) return;
  const int dec = i / L, r = i % L;
  const float4 v = start_torsions(L, seed, decoy0 + dec, r, tors0 ? tors0 + (size_t)dec * L * 3 : nullptr);
  X[i] = v;
  XT[i] = v;
  const ResGeom gi = ideal_geom();  // pose_from_sequence: ideal bond geometry (folding.py:109)
  geom[(size_t)i * 3] = gi.g0; geom[(size_t)i * 3 + 1] = gi.g1; geom[(size_t)i * 3 + 2] = gi.g2;
}
