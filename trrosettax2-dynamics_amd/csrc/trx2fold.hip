// trx2fold.hip -- MI355X (gfx950, wave64) kernels + C ABI of the restraint-guided batched fold.
//
// Replaces the per-decoy PyRosetta process of /root/reference/folding/folding.py (+ folding/utils_ros) by a
// device-resident batched minimiser.  Kernel inventory ("native work-list"):
//   K2 k_build_tables : gen_rst + add_rst selection -> dense (y, y'') spline tables  (utils_ros.py:6-146,706-723)
//   K3/K4 k_pair<BW>  : CB-CB distance / omega / theta / phi spline restraints + soft-sphere repulsion,
//                       energy and Cartesian gradient, lane = decoy                 (folding.py:74-84 score terms)
//   K1/K5/K6 k_chain  : per decoy: gradient slabs -> torsion gradient (suffix scan of force/torque), rama/omega
//                       terms, non-monotone Armijo L-BFGS state machine over the staged protocol, new trial
//                       torsions -> backbone by a parallel rigid-transform scan (NeRF) (folding.py:86-119,164-171)
// Data layout in HBM: see DESIGN.md.  No CPU fallback exists: every entry point runs on the GPU or fails.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/trx2fold.h"
#include "trx2_device.h"

#define KD TRX2_KD
#define KO TRX2_KO
#define KP TRX2_KP
#define LBM TRX2_LBFGS_M
#define CHAIN_THREADS 256
#define PAIR_THREADS 256
#define PAIR_WAVES (PAIR_THREADS / 64)
#define RED_STRIDE 21

// soft-sphere radii: compile-time literals, so the fully unrolled 5 x 5 atom-pair loop of k_pair folds r0^2 and 1/r0^2 into
// its instructions (a runtime table cost 25 scalar loads and 25 IEEE divisions = ~250 vector instructions per kernel)
struct VdwTab { float r0sq[25], ir0sq[25]; };
static constexpr VdwTab make_vdw_tab() {
  constexpr double r0[5][5] = TRX2_VDW_R0_INIT;
  VdwTab t{};
  for (int p = 0; p < 5; p++)
    for (int q = 0; q < 5; q++) {
      t.r0sq[p * 5 + q] = (float)(r0[p][q] * r0[p][q]);
      t.ir0sq[p * 5 + q] = 1.0f / t.r0sq[p * 5 + q];  // correctly rounded, as the device's IEEE division was
    }
  return t;
}
static constexpr VdwTab k_vdw = make_vdw_tab();
__constant__ float c_rama[TRX2_RAMA_NB * 3];  // phi_k, psi_k (rad), p_k
__constant__ float c_rama_sc[TRX2_RAMA_NB * 4];  // sin phi_k, cos phi_k, sin psi_k, cos psi_k
// ideal C-beta placement seen from the ideal local frame: |CA-CB|, angle N-CA-CB, angle C-CA-CB, improper N-C-CA-CB
// (the bonded term's targets; every thread used to rebuild them from the ideal frame at every Cartesian step)
__constant__ float c_cb_ideal[4];

// =================================================================================================
// K2: restraint tables
// =================================================================================================
struct BuildArgs {
  int L, use_orient;
  const float *dist, *omega, *theta, *phi;
  double ebase, erep[3], meff, pcut;
  double bkgr[32];      // background (bins_k/DCUT)^ALPHA per contact bin; computed on the host: a device f64
                        // pow() with a runtime exponent sends the gfx950 backend into a >10 min compile
  const double* knots;  // [107] rounded knot positions: d(35) o(28) t(28) p(16)
  float2 *Td, *To, *Tt, *Tp;
  float *pd, *po, *pt, *pp;
  unsigned char *gen, *sel;
};

__device__ float np_sum_f32_dev(const float* a, int n) {  // numpy pairwise_sum for 8 <= n <= 128
  float r[8];
  _Pragma("unroll 1") for (int j = 0; j < 8; j++) r[j] = a[j];
  int i;
  _Pragma("unroll 1") for (i = 8; i < n - (n % 8); i += 8)
    _Pragma("unroll 1") for (int j = 0; j < 8; j++) r[j] += a[i + j];
  float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  _Pragma("unroll 1") for (; i < n; i++) res += a[i];
  return res;
}

// clamped cubic spline (end slopes 0) second derivatives, then store (y, y'') as float2
__device__ __noinline__ void spline_store(int n, const double* x, const double* y, float2* out) {
  double y2[KD], u[KD];
  y2[0] = -0.5;
  u[0] = (3.0 / (x[1] - x[0])) * ((y[1] - y[0]) / (x[1] - x[0]));
  _Pragma("unroll 1") for (int i = 1; i < n - 1; i++) {
    double sig = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1]);
    double p = sig * y2[i - 1] + 2.0;
    y2[i] = (sig - 1.0) / p;
    double t = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1]);
    u[i] = (6.0 * t / (x[i + 1] - x[i - 1]) - sig * u[i - 1]) / p;
  }
  double un = (3.0 / (x[n - 1] - x[n - 2])) * (0.0 - (y[n - 1] - y[n - 2]) / (x[n - 1] - x[n - 2]));
  y2[n - 1] = (un - 0.5 * u[n - 2]) / (0.5 * y2[n - 2] + 1.0);
  _Pragma("unroll 1") for (int k = n - 2; k >= 0; k--) y2[k] = y2[k] * y2[k + 1] + u[k];
  _Pragma("unroll 1") for (int k = 0; k < n; k++) out[k] = make_float2((float)y[k], (float)y2[k]);
}

__device__ __forceinline__ double round_dp(double v, double scale) { return rint(v * scale) / scale; }

__global__ void k_build_tables(BuildArgs A) {
  const int L = A.L;
  size_t ab = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ab >= (size_t)L * L) return;
  int a = (int)(ab / L), b = (int)(ab % L);
  unsigned char gen = 0, sel = 0;
  const float meff32 = (float)A.meff;
  double y[KD];
  {  // ---- dist (utils_ros.py:54-75)
    const float* row = A.dist + ab * TRX2_ND_BINS;
    float p = np_sum_f32_dev(row + 5, 32);
    A.pd[ab] = p;
    if ((double)p > TRX2_GEN_PCUT && b > a) {
      double attr0 = 0;
      _Pragma("unroll 1") for (int k = 0; k < 32; k++) {
        double bk = A.bkgr[k];  // (bins_k / DCUT)^ALPHA, host libm pow like numpy (utils_ros.py:57)
        float num = row[5 + k] + meff32;
        double den = (double)row[36] * bk + 1e-6;
        double at = -log((double)num / den) + A.ebase;
        if (k == 0) attr0 = at;
        y[3 + k] = round_dp(at, 1e3);
      }
      double rep0 = attr0 > 0.0 ? attr0 : 0.0;
      _Pragma("unroll 1") for (int k = 0; k < 3; k++) y[k] = round_dp(rep0 + A.erep[k], 1e3);
      spline_store(KD, A.knots, y, A.Td + ab * KD);
      gen |= TRX2_M_DIST;
      if ((double)p >= A.pcut) sel |= TRX2_M_DIST;
    }
  }
  if (A.use_orient) {
    _Pragma("unroll 1") for (int ch = 0; ch < 2; ch++) {  // ---- omega, theta (utils_ros.py:81-119), float32 like numpy
      const float* row = (ch == 0 ? A.omega : A.theta) + ab * TRX2_NO_BINS;
      float p = np_sum_f32_dev(row + 1, 24);
      (ch == 0 ? A.po : A.pt)[ab] = p;
      bool ok = (double)p > TRX2_GEN_PCUT && (ch == 0 ? b > a : b != a);
      if (!ok) continue;
      float v[TRX2_NO_BINS];
      float den = row[24] + meff32;
      _Pragma("unroll 1") for (int k = 0; k < TRX2_NO_BINS; k++) v[k] = -(float)log((double)((row[k] + meff32) / den));
      double sc = ch == 0 ? 1e5 : 1e3;
      y[0] = round_dp(v[23], sc);
      y[1] = round_dp(v[24], sc);
      _Pragma("unroll 1") for (int k = 1; k <= 24; k++) y[1 + k] = round_dp(v[k], sc);
      y[26] = round_dp(v[1], sc);
      y[27] = round_dp(v[2], sc);
      spline_store(KO, A.knots + (ch == 0 ? KD : KD + KO), y, (ch == 0 ? A.To : A.Tt) + ab * KO);
      unsigned char bit = ch == 0 ? TRX2_M_OMEGA : TRX2_M_THETA;
      gen |= bit;
      if ((double)p >= A.pcut + 0.5) sel |= bit;
    }
    {  // ---- phi (utils_ros.py:124-144)
      const float* row = A.phi + ab * TRX2_NP_BINS;
      float p = np_sum_f32_dev(row + 1, 12);
      A.pp[ab] = p;
      if ((double)p > TRX2_GEN_PCUT && a != b) {
        float v[TRX2_NP_BINS];
        float den = row[12] + meff32;
        _Pragma("unroll 1") for (int k = 0; k < TRX2_NP_BINS; k++) v[k] = -(float)log((double)((row[k] + meff32) / den));
        y[0] = round_dp(v[2], 1e3);
        y[1] = round_dp(v[1], 1e3);
        _Pragma("unroll 1") for (int k = 1; k <= 12; k++) y[1 + k] = round_dp(v[k], 1e3);
        y[14] = round_dp(v[12], 1e3);
        y[15] = round_dp(v[11], 1e3);
        spline_store(KP, A.knots + KD + 2 * KO, y, A.Tp + ab * KP);
        gen |= TRX2_M_PHI;
        if ((double)p >= A.pcut + 0.6) sel |= TRX2_M_PHI;
      }
    }
  }
  A.gen[ab] = gen;
  A.sel[ab] = sel;
}

// mask2[a][b] = sel[a][b] | sel[b][a] << 4 : both directions of an ordered pair in ONE row-contiguous byte.  k_pair used
// to fetch sel[a][b] and sel[b][a] (a column access: a fresh cache line per visit) before it could even issue its
// coordinate loads -- ~950 of ~3250 cycles per visit (s_memtime stamps, profiles/README.md).
__global__ void k_pack_masks(int L, const unsigned char* sel, unsigned char* mask2) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)L * L) return;
  const int a = (int)(i / L), b = (int)(i % L);
  mask2[i] = (unsigned char)((sel[i] & 15) | ((sel[(size_t)b * L + a] & 15) << 4));
}

// =================================================================================================
// K3/K4: pair terms.  Workgroup = (row residue a, b-range split, decoy group); lane = decoy (BW decoys per
// wave, 64/BW residues b per wave step).  Each ORDERED pair (a,b) is visited from a's row and only the
// gradient on a's atoms is kept -> no atomics, no cross-workgroup reduction, deterministic.
// =================================================================================================
struct PairArgs {
  int L, B, nsplit, Bpad;
  const float4* xyzT;  // [ngrp][L][4][BW] float4 : residue record (N CA C O CB + pad), decoy-minor
  const float2 *Td, *To, *Tt, *Tp;
  const unsigned char* mask;  // [L][L] packed: low nibble = selected bits of (a,b), high nibble = those of (b,a)
  const float* knots;         // [107] float
  const float* wcur;          // [Bpad][8] : w_ap w_dih w_ang w_vdw sep_lo sep_hi active -
  float* fpart;               // [nsplit][Bpad][L][16] gradient on N CA C O CB (+pad)
  float* epart;               // [nsplit][Bpad][L][8]  raw energies dist omega theta phi vdw
  int* seq_ctr;               // evaluation counter in device memory: bumped here, read by the step kernel that follows
};

// ikn[i] = 1 / (kn[i+1] - kn[i]), precomputed once per workgroup: the same correctly rounded quotient the evaluator used
// to compute per term (an IEEE division = ~10 vector instructions, six times per visit)
__device__ __forceinline__ void spline_eval_dev(const float2* __restrict__ row, const float* kn, const float* ikn, int K,
                                                int idx, float x, float& e, float& de) {
  // idx is a guess; fix up against the (rounded, slightly non-uniform) knots
  idx = max(0, min(K - 2, idx));
  if (x < kn[idx]) idx = max(0, idx - 1);
  else if (x >= kn[idx + 1]) idx = min(K - 2, idx + 1);
  float lo = kn[idx], hi = kn[idx + 1];
  float2 k0 = row[idx], k1 = row[idx + 1];
  // the segment's cubic in t = x - lo, formed from (y, y'') of its two knots and evaluated by Horner (17 operations; the
  // symmetric a/b form of the textbook needs ~30):  c1 = (y1-y0)/h - h (2 y0'' + y1'')/6,  c2 = y0''/2,  c3 = (y1''-y0'')/(6h)
  float h = hi - lo, ih = ikn[idx], t = x - lo;
  bool inside = (x > kn[0]) && (x < kn[K - 1]);
  float c1 = fmaf(-h * (1.0f / 6.0f), fmaf(2.0f, k0.y, k1.y), (k1.x - k0.x) * ih);
  float c3 = (k1.y - k0.y) * (ih * (1.0f / 6.0f));
  float ev = fmaf(fmaf(fmaf(c3, t, 0.5f * k0.y), t, c1), t, k0.x);
  float dv = fmaf(fmaf(3.0f * c3, t, k0.y), t, c1);
  // outside the knot range: constant end value, zero slope (SplineFunc).  The end knots are the ones already fetched:
  // x <= kn[0] => idx == 0 => k0 is row[0];  x >= kn[K-1] => idx == K-2 => k1 is row[K-1]
  e = inside ? ev : (x <= kn[0] ? k0.x : k1.x);
  de = inside ? dv : 0.0f;
}

// PAIR_MIN_WAVES (waves per SIMD the register allocator must admit) is a build-time knob so that occupancy-vs-spill
// variants can be A/B-timed on hardware: 2 = no spills (220 VGPRs), 3 = 62 spilled, 4 = 104 spilled (profiles/README.md)
#ifndef PAIR_MIN_WAVES
#define PAIR_MIN_WAVES 2
#endif
// Diagnostic build only (-DTRX2_STAMP, never the shipped library): wave 0 of the workgroup (a = L/2, split 0, group 0)
// accumulates s_memtime cycles per phase; every stamp first drains the memory counters so that a load's latency is charged
// to the phase that issued it.  The drains forbid overlaps the real kernel has: read SHARES, not the total.
#ifdef TRX2_STAMP
__device__ unsigned long long g_stamp[32];
#define STAMP_DECL unsigned long long st_acc[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}; unsigned long long st_prev = 0; \
  const bool st_on = (blockIdx.x == (unsigned)(A.L / 2) && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x >> 6) == 0); \
  if (st_on) { __builtin_amdgcn_s_waitcnt(0); st_prev = __builtin_amdgcn_s_memtime(); }
#define STAMP(k) if (st_on) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
  __builtin_amdgcn_s_waitcnt(0); st_acc[k] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); }
#define STAMP_FLUSH if (st_on && (threadIdx.x & 63) == 0) { for (int k_ = 0; k_ < 16; k_++) g_stamp[k_] = st_acc[k_]; }
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_FLUSH
#endif
// FAM selects the term families an instantiation evaluates: the monolithic kernel (all three) is register-bound at 220
// VGPRs = 2 waves per SIMD (profiles/README.md); each family alone has a much smaller live state.
#define FAM_SYM 1   /* dist + omega: needs CA, CB */
#define FAM_ASYM 2  /* theta + phi (both directions): needs N, CA, CB */
#define FAM_VDW 4   /* soft-sphere repulsion: needs all five atoms, no tables */
#define FAM_ALL 7
template <int BW, int FAM>
__global__ __launch_bounds__(PAIR_THREADS, PAIR_MIN_WAVES) void k_pair(PairArgs A) {
  constexpr int PW = 64 / BW;
  const int L = A.L;
  const int a = blockIdx.x, split = blockIdx.y, grp = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = lane % BW, h = lane / BW;
  const int dec = grp * BW + d;
  const bool live = dec < A.B;

  STAMP_DECL
  __shared__ float s_kn[TRX2_KTOT], s_ikn[TRX2_KTOT];
  __shared__ float s_red[PAIR_WAVES * 64 * RED_STRIDE];  // [slot][decoy][20 (+1 pad: bank-conflict-free)]
  __shared__ unsigned char s_mask[1024];  // packed masks of this workgroup's residues b (chunk <= L <= 1024)
  // One evaluation = one sequence number.  Kept in device memory (not a kernel argument) so that a chunk of
  // (pair, step) launches is a STATIC graph that can be replayed.  The step kernel of this evaluation starts after this
  // kernel has finished (same stream), so every one of its workgroups reads the same, final value.
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && A.seq_ctr) *A.seq_ctr += 1;
  for (int i = threadIdx.x; i < TRX2_KTOT; i += PAIR_THREADS) {
    s_kn[i] = A.knots[i];
    s_ikn[i] = i + 1 < TRX2_KTOT ? 1.0f / (A.knots[i + 1] - A.knots[i]) : 0.0f;  // entries straddling two tables are never read
  }
  {
    const int chunk0 = (L + A.nsplit - 1) / A.nsplit, lo0 = split * chunk0, hi0 = min(L, lo0 + chunk0);
    for (int i = lo0 + threadIdx.x; i < hi0; i += PAIR_THREADS) s_mask[i - lo0] = A.mask[(size_t)a * L + i];
  }
  __syncthreads();
  const float* knd = s_kn;
  const float* kno = s_kn + KD;
  const float* knt = s_kn + KD + KO;
  const float* knp = s_kn + KD + 2 * KO;
  const float *iknd = s_ikn, *ikno = s_ikn + KD, *iknt = s_ikn + KD + KO, *iknp = s_ikn + KD + 2 * KO;
  const float inv_o = 1.0f / (kno[1] - kno[0]), inv_p = 1.0f / (knp[1] - knp[0]);

  float w_ap = 0, w_dih = 0, w_ang = 0, w_vdw = 0;
  int sep_lo = 0, sep_hi = 0;
  bool active = false;
  if (live) {
    const float4* wp = reinterpret_cast<const float4*>(A.wcur + (size_t)dec * 8);
    float4 w0 = wp[0], w1 = wp[1];
    w_ap = w0.x; w_dih = w0.y; w_ang = w0.z; w_vdw = w0.w;
    sep_lo = (int)w1.x; sep_hi = (int)w1.y;
    active = w1.z != 0.0f;
  }

  // residue a
  const float4* xa = A.xyzT + ((size_t)(grp * L + a) * 4) * BW + d;
  float4 q0 = xa[0], q1 = xa[BW], q2 = xa[2 * BW], q3 = xa[3 * BW];
  const f3 Na = mk3(q0.x, q0.y, q0.z), CAa = mk3(q0.w, q1.x, q1.y), Ca = mk3(q1.z, q1.w, q2.x),
           Oa = mk3(q2.y, q2.z, q2.w), CBa = mk3(q3.x, q3.y, q3.z);

  f3 gN = mk3(0, 0, 0), gCA = gN, gC = gN, gO = gN, gCB = gN;
  float e_d = 0, e_o = 0, e_t = 0, e_p = 0, e_v = 0;

  const int chunk = (L + A.nsplit - 1) / A.nsplit;
  const int b_lo = split * chunk, b_hi = min(L, b_lo + chunk);
  STAMP(0)  // prologue: knots to LDS, barrier, weights, residue a

  // The loop runs in blocks of up to 32 visits.  Restraint terms are evaluated in the visit (the pair, hence the table, is
  // the same for all decoys of the wave).  Repulsion is different: WHICH residues touch depends on the decoy, so in lockstep
  // the 25 atom pairs ran whenever ANY of the 64 decoys was within the cutoff -- on ~85 % of the visits of a distance-only
  // fold although ~10 % of (pair, decoy) combinations are in contact (profiles/README.md).  A visit therefore only records
  // a contact bit per lane; after the block every lane walks ITS OWN bits, gathering its own residue b.  The walk takes
  // max-over-lanes(contacts) steps instead of count-of-visits-with-any-contact.  Order per lane stays fixed: deterministic.
  constexpr int VSTRIDE = PAIR_WAVES * PW;
  for (int bb = b_lo + wave * PW; bb < b_hi; bb += 32 * VSTRIDE) {
  unsigned vmask = 0;
#pragma unroll 1
  for (int v = 0; v < 32; v++) {
    const int b0 = bb + v * VSTRIDE;
    if (b0 >= b_hi) break;
    const int b = b0 + h;
    const bool valid = live && active && b < b_hi && b != a;
    const int bc = min(b, L - 1);
    const int sep = abs(a - bc);
    unsigned m_ab = 0, m_ba = 0;
    if (valid && sep >= sep_lo && sep < sep_hi) {
      const unsigned mm = s_mask[bc - b_lo];
      m_ab = mm & 15u;
      m_ba = mm >> 4;
    }
    if (!(FAM & FAM_SYM)) { m_ab &= ~(TRX2_M_DIST | TRX2_M_OMEGA); m_ba &= ~(TRX2_M_DIST | TRX2_M_OMEGA); }
    if (!(FAM & FAM_ASYM)) { m_ab &= ~(TRX2_M_THETA | TRX2_M_PHI); m_ba &= ~(TRX2_M_THETA | TRX2_M_PHI); }
    const unsigned msym = (a < bc) ? m_ab : m_ba;  // DIST / OMEGA bits live on the (min,max) row
    const bool dovdw = (FAM & FAM_VDW) && valid && sep >= TRX2_VDW_MINSEP && w_vdw != 0.0f;
    STAMP(1)  // masks (2 byte loads) + loop control
    if (!__any((int)(m_ab | m_ba | (unsigned)dovdw))) continue;

    const float4* xb = A.xyzT + ((size_t)(grp * L + bc) * 4) * BW + d;
    float4 r0 = xb[0], r1 = xb[BW], r2 = xb[2 * BW], r3 = xb[3 * BW];
    const f3 Nb = mk3(r0.x, r0.y, r0.z), CAb = mk3(r0.w, r1.x, r1.y), Cb = mk3(r1.z, r1.w, r2.x),
             Ob = mk3(r2.y, r2.z, r2.w), CBb = mk3(r3.x, r3.y, r3.z);
    STAMP(2)  // coordinates of residue b (4 x 16 B per lane)
    const size_t iab = (size_t)a * L + bc, iba = (size_t)bc * L + a;
    const size_t isym = (a < bc) ? iab : iba;
    const bool first = a < bc;  // symmetric energies are counted from the lower row only

    if ((FAM & FAM_SYM) && (msym & TRX2_M_DIST)) {
      f3 u = CBa - CBb;
      float d2 = dot(u, u), id = rsqrtf(d2), dd = d2 * id;
      int idx = dd < 2.0f ? 0 : (dd < 3.5f ? 1 : (dd < 4.25f ? 2 : 3 + (int)((dd - 4.25f) * 2.0f)));
      float ev, de;
      spline_eval_dev(A.Td + isym * KD, knd, iknd, KD, idx, dd, ev, de);
      if (first) e_d += ev;
      gCB = fma3(u, w_ap * de * id, gCB);
    }
    STAMP(3)  // dist
    if ((FAM & FAM_SYM) && (msym & TRX2_M_OMEGA)) {
      f3 d1, d2, d3, d4;
      float x = dihedral_grad(CAa, CBa, CBb, CAb, d1, d2, d3, d4);
      float ev, de;
      spline_eval_dev(A.To + isym * KO, kno, ikno, KO, (int)((x - kno[0]) * inv_o), x, ev, de);
      if (first) e_o += ev;
      float s = w_dih * de;
      gCA = fma3(d1, s, gCA);
      gCB = fma3(d2, s, gCB);
    }
    STAMP(4)  // omega
    if ((FAM & FAM_ASYM) && (m_ab & TRX2_M_THETA)) {
      f3 d1, d2, d3, d4;
      float x = dihedral_grad(Na, CAa, CBa, CBb, d1, d2, d3, d4);
      float ev, de;
      spline_eval_dev(A.Tt + iab * KO, knt, iknt, KO, (int)((x - knt[0]) * inv_o), x, ev, de);
      e_t += ev;
      float s = w_dih * de;
      gN = fma3(d1, s, gN);
      gCA = fma3(d2, s, gCA);
      gCB = fma3(d3, s, gCB);
    }
    STAMP(5)  // theta(a,b)
    if ((FAM & FAM_ASYM) && (m_ba & TRX2_M_THETA)) {  // theta(b,a): only its gradient on CB_a (4th point)
      f3 d1, d2, d3, d4;
      float x = dihedral_grad(Nb, CAb, CBb, CBa, d1, d2, d3, d4);
      float ev, de;
      spline_eval_dev(A.Tt + iba * KO, knt, iknt, KO, (int)((x - knt[0]) * inv_o), x, ev, de);
      gCB = fma3(d4, w_dih * de, gCB);
    }
    STAMP(6)  // theta(b,a)
    if ((FAM & FAM_ASYM) && (m_ab & TRX2_M_PHI)) {
      f3 d1, d2, d3;
      float x = angle_grad(CAa, CBa, CBb, d1, d2, d3);
      float ev, de;
      spline_eval_dev(A.Tp + iab * KP, knp, iknp, KP, (int)((x - knp[0]) * inv_p), x, ev, de);
      e_p += ev;
      float s = w_ang * de;
      gCA = fma3(d1, s, gCA);
      gCB = fma3(d2, s, gCB);
    }
    STAMP(7)  // phi(a,b)
    if ((FAM & FAM_ASYM) && (m_ba & TRX2_M_PHI)) {  // phi(b,a): only its gradient on CB_a (3rd point)
      f3 d1, d2, d3;
      float x = angle_grad(CAb, CBb, CBa, d1, d2, d3);
      float ev, de;
      spline_eval_dev(A.Tp + iba * KP, knp, iknp, KP, (int)((x - knp[0]) * inv_p), x, ev, de);
      gCB = fma3(d3, w_ang * de, gCB);
    }
    STAMP(8)  // phi(b,a)
    if ((FAM & FAM_VDW) && dovdw) {
      f3 dca = CAa - CAb;
      if (dot(dca, dca) < (float)TRX2_VDW_CUT2) vmask |= 1u << v;
    }
  }
  while (vmask) {  // per-lane trip count; lanes without further contacts idle
    const int v = __ffs((int)vmask) - 1;
    vmask &= vmask - 1;
    const int b = bb + v * VSTRIDE + h;
    const float4* xb = A.xyzT + ((size_t)(grp * L + b) * 4) * BW + d;
    float4 r0 = xb[0], r1 = xb[BW], r2 = xb[2 * BW], r3 = xb[3 * BW];
    const f3 pa[5] = {Na, CAa, Ca, Oa, CBa};
    const f3 pb[5] = {mk3(r0.x, r0.y, r0.z), mk3(r0.w, r1.x, r1.y), mk3(r1.z, r1.w, r2.x), mk3(r2.y, r2.z, r2.w),
                      mk3(r3.x, r3.y, r3.z)};
    f3 ga[5] = {mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0)};
    float ev = 0;
#pragma unroll
    for (int p = 0; p < 5; p++)
#pragma unroll
      for (int q = 0; q < 5; q++) {
        f3 u = pa[p] - pb[q];
        constexpr VdwTab T = make_vdw_tab();
        const float r02 = T.r0sq[p * 5 + q], ir = T.ir0sq[p * 5 + q];
        float c = fmaxf(r02 - dot(u, u), 0.0f);
        ev = fmaf(c * c, ir, ev);
        ga[p] = fma3(u, -4.0f * c * ir, ga[p]);
      }
    const float s = w_vdw * (float)TRX2_VDW_SCALE;
    if (a < b) e_v += (float)TRX2_VDW_SCALE * ev;  // symmetric energy: counted from the lower row only
    gN = fma3(ga[0], s, gN);
    gCA = fma3(ga[1], s, gCA);
    gC = fma3(ga[2], s, gC);
    gO = fma3(ga[3], s, gO);
    gCB = fma3(ga[4], s, gCB);
  }
  STAMP(9)  // vdw
  }

  STAMP(10) // loop exit
  // ---- reduce over waves and over the PW residue sub-lanes; write decoy-major records.
  // LDS image [slot][decoy][21]: a lane writes its own 20 values at stride 21 (no bank conflict); the readers are
  // (decoy, quad) pairs, 4 lanes per decoy, so every store instruction writes whole 64-B (gradient) / 32-B (energy) runs.
  {
    const int slot = wave * PW + h;
    float* s = s_red + ((size_t)slot * BW + d) * RED_STRIDE;
    const float vals[20] = {gN.x, gN.y, gN.z, gCA.x, gCA.y, gCA.z, gC.x, gC.y, gC.z, gO.x,
                            gO.y, gO.z, gCB.x, gCB.y, gCB.z, e_d, e_o, e_t, e_p, e_v};
#pragma unroll
    for (int k = 0; k < 20; k++) s[k] = vals[k];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 6 * BW; t += PAIR_THREADS) {  // 6 quads per decoy: 4 gradient (16 floats) + 2 energy (8)
    const int dd = t / 6, q = t % 6;
    const int dc = grp * BW + dd;
    if (dc >= A.B) continue;
    float acc[4] = {0, 0, 0, 0};
    const int k0 = q < 4 ? q * 4 : 15 + (q - 4) * 4;  // first value of this quad in the 20-value record
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int k = k0 + i;
      const bool real = q < 4 ? (k < 15) : (k < 20);  // gradient pad (16th float) and energy pads are zero
      if (real)
        for (int sl = 0; sl < PAIR_WAVES * PW; sl++) acc[i] += s_red[((size_t)sl * BW + dd) * RED_STRIDE + k];
    }
    const size_t rec = ((size_t)split * A.Bpad + dc) * L + a;
    const float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (q < 4) reinterpret_cast<float4*>(A.fpart + rec * 16)[q] = v;
    else reinterpret_cast<float4*>(A.epart + rec * 8)[q - 4] = v;
  }
  STAMP(11)  // epilogue: LDS image, barrier, column sums, stores
  STAMP_FLUSH
}

// =================================================================================================
// K1/K5/K6: per-decoy chain kernel
// =================================================================================================
enum { PH_START = 0, PH_LS = 1, PH_DONE = 2 };
enum { MODE_INIT = 0, MODE_STEP = 1, MODE_FINISH = 2 };
// integer state slots
// SI_RUN and SI_SEQ share one aligned 8-byte word: in the fused step launch the two workgroups of a decoy read its state
// while one of them may be writing a run transition; a single 8-byte store / load cannot be seen half-updated, so a
// reader gets (old run, old seq) or (new run, seq of THIS launch -> "already stepped"), never a mixture.
enum { SI_RUN = 0, SI_SEQ, SI_PHASE, SI_ITER, SI_NLS, SI_HL, SI_HH, SI_NH, SI_STATUS, SI_NEVALS, SI_NITERS, SI_N = 16 };
// double state slots
// SD_GAMMA: s.y / y.y of the newest stored pair = the initial Hessian scaling of the two-loop recursion (torsion role)
enum { SD_F = 0, SD_ALPHA, SD_GD, SD_FH0, SD_FH1, SD_FH2, SD_GAMMA, SD_N = 8 };

struct ChainArgs {
  int L, B, Bpad, BW, nsplit, mode, nruns, max_evals;
  const int* seq_ctr;  // evaluation number (device counter bumped by k_pair): a decoy is stepped once per evaluation, by the
                       // torsion OR the Cartesian role (SI_SEQ)
  const trx2_run* runs;
  int* st_i;       // [B][SI_N]
  double* st_d;    // [B][SD_N]
  float* rho;      // [B][LBM]
  float4 *X, *G, *D, *XT;  // [B][L] (phi, psi, omega, -)
  float4 *S, *Y;           // [B][LBM][L]
  float* xyz;              // [B][L][16] trial coordinates, decoy-major
  const float4* geom;      // [B][L][3] internal geometry per residue (ResGeom)
  float4* xyzT;            // decoy-minor copy for k_pair
  float* wcur;             // [Bpad][8]
  const float* fpart;      // [nsplit][Bpad][L][16]
  const float* epart;      // [nsplit][Bpad][L][8]
  double* e_last;          // [B][NTERMS] raw terms of the last evaluation
  double* f_last;          // [B]
  float* grad_out;         // [B][L][3] (MODE_FINISH)
  int* done_count;
};

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
// makes 2 x 12 dependent calls per step.)  Every wave must make the same sequence of calls.
template <int K, int NW>
__device__ __forceinline__ void block_sum_n(double (&v)[K], double* s_buf /* [2][NW*8] */, int& flip) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; k++) v[k] = wave_sum(v[k]);
  if (NW == 1) return;  // the in-wave sum leaves the total in every lane
  double* buf = s_buf + flip * (NW * 8);
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
// L-BFGS history of the decoy staged in LDS for the duration of one step: [LBM][s | y][NT] float4, dynamic shared memory
// (HIST_LDS_BYTES, only the one-residue-per-thread instantiations; gfx950 has 160 KB of LDS per CU).  The two-loop
// recursion is 2 x 12 DEPENDENT rounds; read from global memory every round exposed an L2 round trip (~500-700 of its
// ~1000 cycles; the compiler turns a register prefetch into a wait on the load just issued).  Instead the whole history is
// requested at the top of the step with LDS-DMA loads (global_load_lds_dwordx4: no registers, nothing waits on them until
// the recursion starts a phase later) and every round reads the thread's own slot from LDS.
extern __shared__ float4 s_hist[];
#define HIST_LDS_BYTES(NT) (LBM * 2 * (NT) * 16)
__device__ __forceinline__ void lds_dma16(const float4* src /* per lane */, float4* dst_wave /* wave-uniform: lane i lands at dst + i */) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)dst_wave, 16, 0, 0);
}
// NT threads step one decoy, RPT residues per thread (RPT * NT >= L).  NT = 256 is what runs.  One wave (NT = 64, RPT = 3 at
// L = 150) makes every reduction and scan barrier-free but was SLOWER on MI355X (73.7 vs 61.2 us per evaluation,
// profiles/README.md): the step is bound by the per-thread chain of dependent arithmetic and loads, which RPT multiplies,
// not by its ~25 barriers.
template <int RPT, int NT>
__device__ __forceinline__ void chain_body(const ChainArgs& A, const int dec) {
  constexpr int NW = NT / 64;
  const int L = A.L, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ double s_buf[2 * NW * 8];
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
  if (tid < SI_N && tid >= 2) s_i[tid] = gi[tid];
  if (tid == 0) {  // (run, seq) in ONE 8-byte load
    const unsigned long long rs = *reinterpret_cast<const volatile unsigned long long*>(gi);
    s_i[SI_RUN] = (int)(unsigned)(rs & 0xffffffffull); s_i[SI_SEQ] = (int)(unsigned)(rs >> 32);
  }
  if (tid < SD_N) s_d[tid] = gd_[tid];
  if (tid < LBM) s_rho[tid] = A.rho[(size_t)dec * LBM + tid];
  bsync<NW>();
  int run = s_i[SI_RUN], phase = s_i[SI_PHASE];
  if (A.mode == MODE_STEP && phase == PH_DONE) return;
  const int seq = (A.mode == MODE_STEP) ? *A.seq_ctr : -1;
  if (A.mode == MODE_STEP && (s_i[SI_SEQ] == seq || A.runs[min(run, A.nruns - 1)].cartesian)) return;  // the Cartesian role's turn

  const size_t vb = (size_t)dec * L;  // base of this decoy's [L] vectors
  float4 xt[RPT], gt[RPT];
  bool need_nerf = true;
  constexpr bool HIST_LDS = (RPT == 1);
  if (HIST_LDS && A.mode == MODE_STEP) {
    // the hl stored pairs, newest first; lanes beyond L copy the last residue (no branch around the load), never used
    const int hl0 = s_i[SI_HL], hh0 = s_i[SI_HH], rc = min(tid, L - 1);
    for (int kk = 0; kk < hl0; kk++) {
      const int j = (hh0 - 1 - kk + LBM) % LBM;
      lds_dma16(A.S + ((size_t)dec * LBM + j) * L + rc, s_hist + (j * 2 + 0) * NT + wave * 64);
      lds_dma16(A.Y + ((size_t)dec * LBM + j) * L + rc, s_hist + (j * 2 + 1) * NT + wave * 64);
    }
  }
  CSTAMP(0)  // state load, barrier, role test
  CCOUNT(30)

  if (A.mode != MODE_INIT) {
    // ------------------------------------------------------------------ consume the evaluation at XT
    const trx2_run R = A.runs[min(run, A.nruns - 1)];
    // accepted point, its gradient and the direction: needed only by the state machine below, loaded here so that their
    // latency overlaps the slab loads and the gradient assembly
    float4 x[RPT], g[RPT], dv[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int r = k * NT + tid;
      x[k] = g[k] = dv[k] = make_float4(0, 0, 0, 0);
      if (r < L && A.mode == MODE_STEP) { x[k] = A.X[vb + r]; g[k] = A.G[vb + r]; dv[k] = A.D[vb + r]; }
    }
    double esum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    f3 g2[RPT], g1[RPT];       // per-residue sums of gradient / x cross gradient
    f3 gO_[RPT], gC_[RPT], gCB_[RPT], pN[RPT], pCA[RPT], pC[RPT], pO[RPT], pCB[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int r = k * NT + tid;
      xt[k] = make_float4(0, 0, 0, 0);
      gt[k] = make_float4(0, 0, 0, 0);
      g2[k] = g1[k] = gO_[k] = gC_[k] = gCB_[k] = pN[k] = pCA[k] = pC[k] = pO[k] = pCB[k] = mk3(0, 0, 0);
      if (r < L) {
        xt[k] = A.XT[vb + r];
        float g[16];
#pragma unroll
        for (int i = 0; i < 16; i++) g[i] = 0;
        for (int s = 0; s < A.nsplit; s++) {
          const size_t rec = ((size_t)s * A.Bpad + dec) * L + r;
          const float4* fp = reinterpret_cast<const float4*>(A.fpart + rec * 16);
#pragma unroll
          for (int q = 0; q < 4; q++) {
            float4 v = fp[q];
            g[q * 4] += v.x; g[q * 4 + 1] += v.y; g[q * 4 + 2] += v.z; g[q * 4 + 3] += v.w;
          }
          const float4* ep = reinterpret_cast<const float4*>(A.epart + rec * 8);
          float4 e0 = ep[0], e1 = ep[1];
          esum[0] += e0.x; esum[1] += e0.y; esum[2] += e0.z; esum[3] += e0.w; esum[4] += e1.x;
        }
        const float4* xp = reinterpret_cast<const float4*>(A.xyz + (vb + r) * 16);
        float4 c0 = xp[0], c1 = xp[1], c2 = xp[2], c3 = xp[3];
        pN[k] = mk3(c0.x, c0.y, c0.z); pCA[k] = mk3(c0.w, c1.x, c1.y); pC[k] = mk3(c1.z, c1.w, c2.x);
        pO[k] = mk3(c2.y, c2.z, c2.w); pCB[k] = mk3(c3.x, c3.y, c3.z);
        f3 gN = mk3(g[0], g[1], g[2]), gCA = mk3(g[3], g[4], g[5]);
        gC_[k] = mk3(g[6], g[7], g[8]); gO_[k] = mk3(g[9], g[10], g[11]); gCB_[k] = mk3(g[12], g[13], g[14]);
        g2[k] = gN + gCA + gC_[k] + gO_[k] + gCB_[k];
        g1[k] = cross(pN[k], gN) + cross(pCA[k], gCA) + cross(pC[k], gC_[k]) + cross(pO[k], gO_[k]) + cross(pCB[k], gCB_[k]);
        // torsion-space terms: rama (residues 2..L-1) and omega_bb (1..L-1)
        if (r >= 1 && r < L - 1) {
          float s = 0, dph = 0, dps = 0;
          // sin/cos of (phi - phi_k), (psi - psi_k) by the angle-addition identities: 2 sincosf per residue, not 12
          float sph, cph, sps, cps;
          fast_sincosf(xt[k].x, &sph, &cph);
          fast_sincosf(xt[k].y, &sps, &cps);
#pragma unroll
          for (int j = 0; j < TRX2_RAMA_NB; j++) {
            const float sk = c_rama_sc[j * 4], ck = c_rama_sc[j * 4 + 1], tk = c_rama_sc[j * 4 + 2], uk = c_rama_sc[j * 4 + 3];
            const float sa = sph * ck - cph * sk, ca = cph * ck + sph * sk;
            const float sb = sps * uk - cps * tk, cb = cps * uk + sps * tk;
            float t = c_rama[j * 3 + 2] * expf((float)TRX2_RAMA_KAPPA * (ca + cb - 2.0f));
            s += t; dph -= t * (float)TRX2_RAMA_KAPPA * sa; dps -= t * (float)TRX2_RAMA_KAPPA * sb;
          }
          float inv = 1.0f / (s + (float)TRX2_RAMA_FLOOR);
          esum[5] += -(double)logf((s + (float)TRX2_RAMA_FLOOR) * (1.0f / (float)TRX2_RAMA_PREF));
          gt[k].x += -R.w[4] * dph * inv;
          gt[k].y += -R.w[4] * dps * inv;
        }
        if (r < L - 1) {
          float dw = xt[k].z - TRX2_PI_F;
          dw -= 2.0f * TRX2_PI_F * rintf(dw * (0.5f / TRX2_PI_F));
          dw *= (1.0f / TRX2_DEG_F);
          esum[6] += (double)((float)TRX2_OMEGA_K * dw * dw);
          gt[k].z += R.w[5] * 2.0f * (float)TRX2_OMEGA_K * dw * (1.0f / TRX2_DEG_F);
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
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
        for (int i = 0; i < 6; i++) {
          float t = __shfl_down(v[i], o, 64);
          if (lane + o < 64) v[i] += t;
        }
      }
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
          const float* nx = A.xyz + (vb + r + 1) * 16;
          f3 Nn = mk3(nx[0], nx[1], nx[2]);
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
    block_sum_n<8, NW>(esum, s_buf, flip);
    const double f_t = (double)R.w[0] * esum[0] + (double)R.w[1] * (esum[1] + esum[2]) + (double)R.w[2] * esum[3] +
                       (double)R.w[3] * esum[4] + (double)R.w[4] * esum[5] + (double)R.w[5] * esum[6];
    if (tid < TRX2_NTERMS) A.e_last[(size_t)dec * TRX2_NTERMS + tid] = esum[tid];
    if (tid == 0) A.f_last[dec] = f_t;

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
    bool next_run = false, new_dir = false, steepest = false, new_trial = false;
    const bool finite_t = isfinite(f_t);
    CSTAMP(3)  // energy reduction + loads of X, G, D
    if (!finite_t && phase == PH_START) { status = TRX2_DIVERGED; phase = PH_DONE; }
    else if (phase == PH_START) {
      if (R.precheck && esum[5] + esum[4] < (double)TRX2_CLASH_BREAK) {
        run = R.skip_to;
        if (run >= A.nruns) phase = PH_DONE;
        need_nerf = false;
      } else {
        f = f_t;
#pragma unroll
        for (int k = 0; k < RPT; k++) g[k] = gt[k];
        hl = 0; hh = 0; nh = 1; fh[0] = f; iter = 0;
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
          v3[0] += (double)dot3(s[k], y[k]); v3[1] += (double)dot3(s[k], s[k]); v3[2] += (double)dot3(y[k], y[k]);
        }
        block_sum_n<3, NW>(v3, s_buf, flip);
        if (v3[0] > 1e-12 * sqrt(v3[1] * v3[2])) {
#pragma unroll
          for (int k = 0; k < RPT; k++) {
            const int r = k * NT + tid;
            if (r < L) {
              A.S[((size_t)dec * LBM + hh) * L + r] = s[k];
              A.Y[((size_t)dec * LBM + hh) * L + r] = y[k];
            }
          }
          if (HIST_LDS) {  // slot hh of the staged copy: its DMA (the oldest pair) must have landed before it is replaced
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_hist[(hh * 2 + 0) * NT + tid] = s[0];
            s_hist[(hh * 2 + 1) * NT + tid] = y[0];
          }
          bsync<NW>();
          if (tid == 0) s_rho[hh] = (float)(1.0 / v3[0]);
          gamma_h = v3[0] / v3[2];
          bsync<NW>();
          hh = (hh + 1) % LBM;
          if (hl < LBM) hl++;
        }
        const double fprev = f;
#pragma unroll
        for (int k = 0; k < RPT; k++) { x[k] = xt[k]; g[k] = gt[k]; }
        f = f_t;
        if (nh < TRX2_LS_PAST) fh[nh++] = f;
        else { fh[0] = fh[1]; fh[1] = fh[2]; fh[2] = f; }
        iter++; n_iters++;
        const bool conv = 2.0 * fabs(fprev - f) <= (double)TRX2_MIN_TOL * (fabs(fprev) + fabs(f) + 1e-10);
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
      {
        float4 q[RPT];
#pragma unroll
        for (int k = 0; k < RPT; k++) q[k] = g[k];
        if (HIST_LDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the staged history has landed (long ago)
        auto load_pair = [&](int j, float4 (&s_)[RPT], float4 (&y_)[RPT]) {
          if (HIST_LDS) {
            const float4 z = make_float4(0, 0, 0, 0), sv = s_hist[(j * 2 + 0) * NT + tid], yv = s_hist[(j * 2 + 1) * NT + tid];
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
        alpha = fmin(1.0, 1.0 / sqrt(v1[0]));
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
    // ---- store state
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int r = k * NT + tid;
      if (r < L) { A.X[vb + r] = x[k]; A.G[vb + r] = g[k]; A.D[vb + r] = dv[k]; A.XT[vb + r] = xt[k]; }
    }
    bsync<NW>();
    if (tid == 0) {
      gi[SI_PHASE] = phase; gi[SI_ITER] = iter; gi[SI_NLS] = nls; gi[SI_HL] = hl; gi[SI_HH] = hh;
      gi[SI_NH] = nh; gi[SI_STATUS] = status; gi[SI_NEVALS] = n_evals; gi[SI_NITERS] = n_iters;
      *reinterpret_cast<volatile unsigned long long*>(gi) = ((unsigned long long)(unsigned)seq << 32) | (unsigned long long)(unsigned)run;  // last, in one piece
      gd_[SD_F] = f; gd_[SD_ALPHA] = alpha; gd_[SD_GD] = gdir; gd_[SD_FH0] = fh[0]; gd_[SD_FH1] = fh[1]; gd_[SD_FH2] = fh[2];
      gd_[SD_GAMMA] = gamma_h;
      if (phase == PH_DONE) atomicAdd(A.done_count, 1);
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
  // ------------------------------------------------------------------ weights for the next pair launch
  if (tid == 0) {
    const trx2_run Rn = A.runs[min(run, A.nruns - 1)];
    float* w = A.wcur + (size_t)dec * 8;
    w[0] = Rn.w[0]; w[1] = Rn.w[1]; w[2] = Rn.w[2]; w[3] = Rn.w[3];
    w[4] = (float)Rn.sep_lo; w[5] = (float)Rn.sep_hi;
    w[6] = (phase == PH_DONE && A.mode == MODE_STEP) ? 0.0f : 1.0f;
    w[7] = 0;
  }
  if (!need_nerf) return;

  // ------------------------------------------------------------------ K1: torsions XT -> backbone (NeRF scan)
  // M_r maps frame r+1 coordinates into frame r; F_r = F_0 o M_0 o ... o M_{r-1}
  bsync<NW>();
#pragma unroll
  for (int k = 0; k < RPT; k++) s_phi[k * NT + tid] = xt[k].x;
  bsync<NW>();
  const float4* gq = A.geom + vb * 3;
  Xf carry;  // F_0: N at the origin, CA on +x, C in the xy plane (same start as the oracle)
  {
    const float4 q0 = gq[0];
    float sa, ca;
    fast_sincosf(q0.w, &sa, &ca);
    carry = xf_from_atoms(mk3(0, 0, 0), mk3(q0.x, 0, 0), mk3(q0.x - q0.y * ca, q0.y * sa, 0));
  }
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int r = k * NT + tid;
    Xf M = xf_identity();
    ResGeom gr = ideal_geom();
    f3 lN = mk3(0, 0, 0), lCA = lN, lC = lN, lCB = lN;
    if (r < L) {
      gr.g0 = gq[r * 3]; gr.g1 = gq[r * 3 + 1]; gr.g2 = gq[r * 3 + 2];
      local_atoms(gr, lN, lCA, lC, lCB);
      if (r + 1 < L) {
        const float4 n0 = gq[(r + 1) * 3];  // next residue: |N-CA|, |CA-C|, angle N-CA-C
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
      }
    }
    CSTAMP(10)  // NeRF: geometry loads, sincos, local frames
    // inclusive scan of M within the wave
    Xf P = M;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      Xf t = xf_shfl_up(P, o);
      if (lane >= o) P = xf_compose(t, P);
    }
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
    Xf prev = xf_shfl_up(P, 1);
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
      float4* xo = reinterpret_cast<float4*>(A.xyz + (vb + r) * 16);
      xo[0] = o0; xo[1] = o1; xo[2] = o2; xo[3] = o3;
      const int grp = dec / A.BW, dd = dec % A.BW;
      float4* xT = A.xyzT + ((size_t)(grp * L + r) * 4) * A.BW + dd;
      xT[0] = o0; xT[A.BW] = o1; xT[2 * A.BW] = o2; xT[3 * A.BW] = o3;
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
  int L, B, Bpad, BW, nsplit, nruns, max_evals;
  const int* seq_ctr;
  const trx2_run* runs;
  int* st_i; double* st_d; float* rho;
  float4 *CX, *CG, *CD;      // [B][L][4] accepted point, its gradient, direction
  float4 *CS, *CY;           // [B][LBM][L][4]
  float* xyz;                // [B][L][16] trial coordinates = trial DOF vector (in/out)
  float4* xyzT;
  float4 *X, *XT, *geom;     // torsions and internal geometry, written when the run ends
  float* wcur;
  const float* fpart; const float* epart;
  double *e_last, *f_last;
  int* done_count;
};
__device__ __forceinline__ float dot4(float4 a, float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }
struct Res5 { f3 N, CA, C, O, CB; };
__device__ __forceinline__ Res5 unpack5(const float* p) {
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

template <int NT>
__device__ __forceinline__ void cart_body(const CartArgs& A, const int dec) {
  constexpr int NW = NT / 64;  // one residue per thread: NT = 256 for chains up to 256 residues, 512 up to 512
  const int L = A.L, tid = threadIdx.x, r = tid;
  const bool act = r < L;
  __shared__ double s_buf[2 * NW * 8];
  int flip = 0;
  __shared__ float s_alpha[LBM];
  __shared__ int s_i[SI_N];
  __shared__ double s_d[SD_N];
  __shared__ float s_rho[LBM];
  __shared__ float s_xyz[NT * 16];
  __shared__ float s_dt[NT * 3];
  int* gi = A.st_i + (size_t)dec * SI_N;
  double* gd_ = A.st_d + (size_t)dec * SD_N;
  KSTAMP_DECL
  if (tid < SI_N && tid >= 2) s_i[tid] = gi[tid];
  if (tid == 0) {  // (run, seq) in ONE 8-byte load
    const unsigned long long rs = *reinterpret_cast<const volatile unsigned long long*>(gi);
    s_i[SI_RUN] = (int)(unsigned)(rs & 0xffffffffull); s_i[SI_SEQ] = (int)(unsigned)(rs >> 32);
  }
  if (tid < SD_N) s_d[tid] = gd_[tid];
  if (tid < LBM) s_rho[tid] = A.rho[(size_t)dec * LBM + tid];
  __syncthreads();
  int run = s_i[SI_RUN], phase = s_i[SI_PHASE];
  const int seq = *A.seq_ctr;
  if (phase == PH_DONE || s_i[SI_SEQ] == seq) return;
  const trx2_run R = A.runs[min(run, A.nruns - 1)];
  if (!R.cartesian) return;
  const size_t vb = (size_t)dec * L;
  CSTAMP(16)  // state load, role test
  CCOUNT(28)

  // ---- trial coordinates; neighbours through LDS
  float4 xt[4], gt[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { xt[q] = gt[q] = make_float4(0, 0, 0, 0); }
  double esum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (act) {
    const float4* xp = reinterpret_cast<const float4*>(A.xyz + (vb + r) * 16);
#pragma unroll
    for (int q = 0; q < 4; q++) { xt[q] = xp[q]; reinterpret_cast<float4*>(s_xyz + r * 16)[q] = xt[q]; }
    for (int s = 0; s < A.nsplit; s++) {
      const size_t rec = ((size_t)s * A.Bpad + dec) * L + r;
      const float4* fp = reinterpret_cast<const float4*>(A.fpart + rec * 16);
#pragma unroll
      for (int q = 0; q < 4; q++) { float4 v = fp[q]; gt[q].x += v.x; gt[q].y += v.y; gt[q].z += v.z; gt[q].w += v.w; }
      const float4* ep = reinterpret_cast<const float4*>(A.epart + rec * 8);
      float4 e0 = ep[0], e1 = ep[1];
      esum[0] += e0.x; esum[1] += e0.y; esum[2] += e0.z; esum[3] += e0.w; esum[4] += e1.x;
    }
    gt[3].w = 0.0f;
  }
  __syncthreads();
  CSTAMP(17)  // coordinates -> LDS, slab sums
  f3 aN = mk3(0, 0, 0), aCA = aN, aC = aN, aO = aN, aCB = aN;  // gradient of the local terms on this residue's atoms
  float dphi = 0, dpsi = 0, dom = 0;
  Res5 Me = unpack5(s_xyz + (act ? r : 0) * 16), Pv = Me, Nx = Me;
  if (act && r > 0) Pv = unpack5(s_xyz + (r - 1) * 16);
  if (act && r + 1 < L) Nx = unpack5(s_xyz + (r + 1) * 16);
  if (act) {
    // rama (residues 2..L-1) and omega_bb (1..L-1): derivatives with respect to the torsion angles
    f3 t1, t2, t3, t4;
    if (r >= 1 && r < L - 1) {
      const float ph = dihedral_grad(Pv.C, Me.N, Me.CA, Me.C, t1, t2, t3, t4), ps = dihedral_grad(Me.N, Me.CA, Me.C, Nx.N, t1, t2, t3, t4);
      float sph, cph, sps, cps, sm = 0, a1 = 0, a2 = 0;
      fast_sincosf(ph, &sph, &cph); fast_sincosf(ps, &sps, &cps);
#pragma unroll
      for (int j = 0; j < TRX2_RAMA_NB; j++) {
        const float sk = c_rama_sc[j * 4], ck = c_rama_sc[j * 4 + 1], tk = c_rama_sc[j * 4 + 2], uk = c_rama_sc[j * 4 + 3];
        const float sa = sph * ck - cph * sk, ca = cph * ck + sph * sk, sb = sps * uk - cps * tk, cb = cps * uk + sps * tk;
        const float t = c_rama[j * 3 + 2] * expf((float)TRX2_RAMA_KAPPA * (ca + cb - 2.0f));
        sm += t; a1 -= t * (float)TRX2_RAMA_KAPPA * sa; a2 -= t * (float)TRX2_RAMA_KAPPA * sb;
      }
      const float inv = 1.0f / (sm + (float)TRX2_RAMA_FLOOR);
      esum[5] += -(double)logf((sm + (float)TRX2_RAMA_FLOOR) * (1.0f / (float)TRX2_RAMA_PREF));
      dphi = -R.w[4] * a1 * inv; dpsi = -R.w[4] * a2 * inv;
    }
    if (r < L - 1) {
      float dw = wrap_pi_f(dihedral_grad(Me.CA, Me.C, Nx.N, Nx.CA, t1, t2, t3, t4) - TRX2_PI_F) * (1.0f / TRX2_DEG_F);
      esum[6] += (double)((float)TRX2_OMEGA_K * dw * dw);
      dom = R.w[5] * 2.0f * (float)TRX2_OMEGA_K * dw * (1.0f / TRX2_DEG_F);
    }
    s_dt[r * 3] = dphi; s_dt[r * 3 + 1] = dpsi; s_dt[r * 3 + 2] = dom;
  }
  __syncthreads();
  CSTAMP(18)  // rama / omega: angles and dE/dangle
  if (act) {
    f3 d1, d2, d3, d4;
    if (dphi != 0.0f) { dihedral_grad(Pv.C, Me.N, Me.CA, Me.C, d1, d2, d3, d4); aN = fma3(d2, dphi, aN); aCA = fma3(d3, dphi, aCA); aC = fma3(d4, dphi, aC); }
    if (dpsi != 0.0f) { dihedral_grad(Me.N, Me.CA, Me.C, Nx.N, d1, d2, d3, d4); aN = fma3(d1, dpsi, aN); aCA = fma3(d2, dpsi, aCA); aC = fma3(d3, dpsi, aC); }
    if (dom != 0.0f) { dihedral_grad(Me.CA, Me.C, Nx.N, Nx.CA, d1, d2, d3, d4); aCA = fma3(d1, dom, aCA); aC = fma3(d2, dom, aC); }
    if (r + 1 < L) {  // phi of the next residue moves C of this one
      const float c = s_dt[(r + 1) * 3];
      if (c != 0.0f) { Res5 N2 = Nx; dihedral_grad(Me.C, N2.N, N2.CA, N2.C, d1, d2, d3, d4); aC = fma3(d1, c, aC); }
    }
    if (r > 0) {      // psi and omega of the previous residue move N (and CA) of this one
      const float c1 = s_dt[(r - 1) * 3 + 1], c2 = s_dt[(r - 1) * 3 + 2];
      if (c1 != 0.0f) { dihedral_grad(Pv.N, Pv.CA, Pv.C, Me.N, d1, d2, d3, d4); aN = fma3(d4, c1, aN); }
      if (c2 != 0.0f) { dihedral_grad(Pv.CA, Pv.C, Me.N, Me.CA, d1, d2, d3, d4); aN = fma3(d3, c2, aN); aCA = fma3(d4, c2, aCA); }
    }
    // bonded term: ideal CB geometry from the ideal local frame
    CSTAMP(19)  // rama / omega gradients on atoms (up to 6 dihedral gradients)
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
      if (r + 1 < L) { LinkGrad G = link_terms(Me, Nx); eb += G.e; bCA += G.CA; bC += G.C; bO += G.O; }
      if (r > 0) { LinkGrad G = link_terms(Pv, Me); bN += G.Nn; bCA += G.CAn; }
      esum[7] += (double)eb;
      aN = fma3(bN, wcb, aN); aCA = fma3(bCA, wcb, aCA); aC = fma3(bC, wcb, aC); aO = fma3(bO, wcb, aO); aCB = fma3(bCB, wcb, aCB);
    }
    gt[0].x += aN.x; gt[0].y += aN.y; gt[0].z += aN.z; gt[0].w += aCA.x;
    gt[1].x += aCA.y; gt[1].y += aCA.z; gt[1].z += aC.x; gt[1].w += aC.y;
    gt[2].x += aC.z; gt[2].y += aO.x; gt[2].z += aO.y; gt[2].w += aO.z;
    gt[3].x += aCB.x; gt[3].y += aCB.y; gt[3].z += aCB.z;
  }
  CSTAMP(20)  // bonded term
  block_sum_n<8, NW>(esum, s_buf, flip);
  const double f_t = (double)R.w[0] * esum[0] + (double)R.w[1] * (esum[1] + esum[2]) + (double)R.w[2] * esum[3] + (double)R.w[3] * esum[4] +
                     (double)R.w[4] * esum[5] + (double)R.w[5] * esum[6] + (double)R.w[6] * esum[7];
  if (tid < TRX2_NTERMS) A.e_last[(size_t)dec * TRX2_NTERMS + tid] = esum[tid];
  if (tid == 0) A.f_last[dec] = f_t;

  // ------------------------------------------------------------------ minimiser state machine (as k_chain, 4 float4 per residue)
  int iter = s_i[SI_ITER], nls = s_i[SI_NLS], hl = s_i[SI_HL], hh = s_i[SI_HH], nh = s_i[SI_NH];
  int n_evals = s_i[SI_NEVALS] + 1, n_iters = s_i[SI_NITERS], status = s_i[SI_STATUS];
  double f = s_d[SD_F], alpha = s_d[SD_ALPHA], gdir = s_d[SD_GD];
  double fh[3] = {s_d[SD_FH0], s_d[SD_FH1], s_d[SD_FH2]};
  double gamma_h = s_d[SD_GAMMA];
  float4 x[4], g[4], dv[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    x[q] = g[q] = dv[q] = make_float4(0, 0, 0, 0);
    if (act) { x[q] = A.CX[(vb + r) * 4 + q]; g[q] = A.CG[(vb + r) * 4 + q]; dv[q] = A.CD[(vb + r) * 4 + q]; }
  }
  bool next_run = false, new_dir = false, steepest = false, new_trial = false;
  const bool finite_t = isfinite(f_t);
  CSTAMP(21)  // energy reduction, loads of X, G, D
  if (!finite_t && phase == PH_START) { status = TRX2_DIVERGED; phase = PH_DONE; }
  else if (phase == PH_START) {
    f = f_t;
#pragma unroll
    for (int q = 0; q < 4; q++) { x[q] = xt[q]; g[q] = gt[q]; }
    hl = 0; hh = 0; nh = 1; fh[0] = f; iter = 0;
    steepest = true;
  } else {
    double fref = fh[0];
    for (int k = 1; k < nh; k++) fref = fmax(fref, fh[k]);
    const bool accept = finite_t && f_t <= fref + (double)TRX2_LS_C1 * alpha * gdir;
    if (accept) {
      double v3[3] = {0, 0, 0};
      float4 sv[4], yv[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        sv[q] = make_float4(xt[q].x - x[q].x, xt[q].y - x[q].y, xt[q].z - x[q].z, xt[q].w - x[q].w);
        yv[q] = make_float4(gt[q].x - g[q].x, gt[q].y - g[q].y, gt[q].z - g[q].z, gt[q].w - g[q].w);
        v3[0] += (double)dot4(sv[q], yv[q]); v3[1] += (double)dot4(sv[q], sv[q]); v3[2] += (double)dot4(yv[q], yv[q]);
      }
      block_sum_n<3, NW>(v3, s_buf, flip);
      if (v3[0] > 1e-12 * sqrt(v3[1] * v3[2])) {
        if (act)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            A.CS[(((size_t)dec * LBM + hh) * L + r) * 4 + q] = sv[q];
            A.CY[(((size_t)dec * LBM + hh) * L + r) * 4 + q] = yv[q];
          }
        __syncthreads();
        if (tid == 0) s_rho[hh] = (float)(1.0 / v3[0]);
        gamma_h = v3[0] / v3[2];
        __syncthreads();
        hh = (hh + 1) % LBM;
        if (hl < LBM) hl++;
      }
      const double fprev = f;
#pragma unroll
      for (int q = 0; q < 4; q++) { x[q] = xt[q]; g[q] = gt[q]; }
      f = f_t;
      if (nh < TRX2_LS_PAST) fh[nh++] = f;
      else { fh[0] = fh[1]; fh[1] = fh[2]; fh[2] = f; }
      iter++; n_iters++;
      const bool conv = 2.0 * fabs(fprev - f) <= (double)TRX2_MIN_TOL * (fabs(fprev) + fabs(f) + 1e-10);
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
    // Two-loop recursion, 8 float4 per thread and stored pair.  Loads are branch-free (index clamped to the last residue; idle
    // threads are masked out of the dot and the update instead) so that a pair's eight loads issue together -- guarded
    // per element, each load had its own branch and wait (~2000 cycles per round, 57 % of a Cartesian step: s_memtime
    // stamps, profiles/README.md).  Two named buffers and a loop unrolled by two keep the next pair in flight without a
    // register copy (a copy makes the compiler wait for the load it has just issued).
    float4 qv[4];
#pragma unroll
    for (int q = 0; q < 4; q++) qv[q] = g[q];
    const int rc = min(r, L - 1);
    auto pair_at = [&](int kk) { return (hh - 1 - kk + LBM) % LBM; };
    auto load_pair = [&](int kk, float4 (&s_)[4], float4 (&y_)[4]) {
      const size_t o = (((size_t)dec * LBM + pair_at(kk)) * L + rc) * 4;
#pragma unroll
      for (int q = 0; q < 4; q++) { s_[q] = A.CS[o + q]; y_[q] = A.CY[o + q]; }
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
    float4 sa[4], ya[4], sb[4], yb[4];
    if (hl > 0) load_pair(0, sa, ya);
    for (int kk = 0; kk < hl; kk += 2) {  // pair kk in buffer a, pair kk+1 in buffer b
      if (kk + 1 < hl) load_pair(kk + 1, sb, yb);
      round1(kk, sa, ya);
      if (kk + 1 < hl) {
        if (kk + 2 < hl) load_pair(kk + 2, sa, ya);
        round1(kk + 1, sb, yb);
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
    // backwards: the oldest pair (hl-1) is still in its buffer -- a if hl is odd, b if even
    if (hl > 0) {
      int kk = hl - 1;
      if (kk & 1) {  // pair kk sits in b
        if (kk > 0) load_pair(kk - 1, sa, ya);
        round2(kk, sb, yb);
        kk--;
      }
      for (; kk >= 0; kk -= 2) {  // pair kk in a, pair kk-1 goes to b
        if (kk > 0) load_pair(kk - 1, sb, yb);
        round2(kk, sa, ya);
        if (kk > 0) {
          if (kk > 1) load_pair(kk - 2, sa, ya);
          round2(kk - 1, sb, yb);
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
  if (steepest) {
    double v1[1] = {0};
#pragma unroll
    for (int q = 0; q < 4; q++) { dv[q] = make_float4(-g[q].x, -g[q].y, -g[q].z, -g[q].w); v1[0] += (double)dot4(g[q], g[q]); }
    block_sum_n<1, NW>(v1, s_buf, flip);
    if (!(v1[0] > 0)) next_run = true;
    else { gdir = -v1[0]; alpha = fmin(1.0, 1.0 / sqrt(v1[0])); nls = 0; new_trial = true; }
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
  // ---- store state and the coordinates for the next pair launch
  if (act) {
#pragma unroll
    for (int q = 0; q < 4; q++) { A.CX[(vb + r) * 4 + q] = x[q]; A.CG[(vb + r) * 4 + q] = g[q]; A.CD[(vb + r) * 4 + q] = dv[q]; }
    float4* xo = reinterpret_cast<float4*>(A.xyz + (vb + r) * 16);
    const int grp = dec / A.BW, dd = dec % A.BW;
    float4* xT = A.xyzT + ((size_t)(grp * L + r) * 4) * A.BW + dd;
#pragma unroll
    for (int q = 0; q < 4; q++) { xo[q] = xt[q]; xT[q * A.BW] = xt[q]; }
  }
  CSTAMP(26)  // direction test, trial point, state + coordinate stores
  // ---- leaving Cartesian space (run finished, or the decoy stops here on its evaluation budget / divergence): torsions +
  //      relaxed internal geometry of the ACCEPTED point, for the torsion-space runs after it and for the final report
  if (next_run || phase == PH_DONE) {
    __syncthreads();
    if (act)
#pragma unroll
      for (int q = 0; q < 4; q++) reinterpret_cast<float4*>(s_xyz + r * 16)[q] = x[q];
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
  if (tid == 0) {
    gi[SI_PHASE] = phase; gi[SI_ITER] = iter; gi[SI_NLS] = nls; gi[SI_HL] = hl; gi[SI_HH] = hh;
    gi[SI_NH] = nh; gi[SI_STATUS] = status; gi[SI_NEVALS] = n_evals; gi[SI_NITERS] = n_iters;
    *reinterpret_cast<volatile unsigned long long*>(gi) = ((unsigned long long)(unsigned)seq << 32) | (unsigned long long)(unsigned)run;  // last, in one piece
    gd_[SD_F] = f; gd_[SD_ALPHA] = alpha; gd_[SD_GD] = gdir; gd_[SD_FH0] = fh[0]; gd_[SD_FH1] = fh[1]; gd_[SD_FH2] = fh[2];
    gd_[SD_GAMMA] = gamma_h;
    if (phase == PH_DONE) atomicAdd(A.done_count, 1);
    const trx2_run Rn = A.runs[min(run, A.nruns - 1)];
    float* w = A.wcur + (size_t)dec * 8;
    w[0] = Rn.w[0]; w[1] = Rn.w[1]; w[2] = Rn.w[2]; w[3] = Rn.w[3];
    w[4] = (float)Rn.sep_lo; w[5] = (float)Rn.sep_hi; w[6] = (phase == PH_DONE) ? 0.0f : 1.0f; w[7] = 0;
  }
  if (tid < LBM) A.rho[(size_t)dec * LBM + tid] = s_rho[tid];
}

// ---- launchable forms.  k_chain: INIT / FINISH passes and protocols without a Cartesian run.  k_step: one launch of 2B
// workgroups per evaluation -- workgroup d < B steps decoy d in torsion space, workgroup B + d steps it in Cartesian
// space; whichever does not match the decoy's current run exits at once.  The two roles touch disjoint decoys, so they
// run concurrently instead of as two half-empty launches back to back (k_cart alone was 22-27 % of GPU time).
// The torsion role runs on TN threads, RPT residues each.
template <int RPT, int TN>
__global__ __launch_bounds__(TN) void k_chain(ChainArgs A) { chain_body<RPT, TN>(A, blockIdx.x); }
template <int RPT, int TN, int NT>
__global__ __launch_bounds__(NT) void k_step(ChainArgs A, CartArgs C) {
  if ((int)blockIdx.x < A.B) {
    if (NT == TN || threadIdx.x < TN) chain_body<RPT, TN>(A, blockIdx.x);  // the other waves of the workgroup exit at once
  } else cart_body<NT>(C, (int)blockIdx.x - A.B);
}

// random start torsions: set_random_dihedral (utils_ros.py:656-696) with explicit (seed, decoy, residue) hashing
__device__ __forceinline__ uint64_t splitmix64_dev(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ void k_init_torsions(int L, int B, uint64_t seed, uint32_t decoy0, const float* tors0, float4* X, float4* XT, float4* geom) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L * B) return;
  const int dec = i / L, r = i % L;
  float4 v;
  if (tors0) v = make_float4(tors0[(size_t)i * 3], tors0[(size_t)i * 3 + 1], tors0[(size_t)i * 3 + 2], 0);
  else {
    float ph = 180.0f, ps = 180.0f;
    if (r < L - 1) {
      const double cum[6] = TRX2_RAND_CUM_INIT;
      uint64_t hsh = splitmix64_dev(seed ^ splitmix64_dev(((uint64_t)(decoy0 + dec) << 32) | (uint32_t)r));
      double u = (double)(hsh >> 11) * (1.0 / 9007199254740992.0);
      int k = 0;
      while (!(u <= cum[k])) k++;
      v = make_float4(c_rama[k * 3], c_rama[k * 3 + 1], TRX2_PI_F, 0);
    } else
      v = make_float4(ph * TRX2_DEG_F, ps * TRX2_DEG_F, TRX2_PI_F, 0);
  }
  X[i] = v;
  XT[i] = v;
  const ResGeom gi = ideal_geom();  // pose_from_sequence: ideal bond geometry (folding.py:109)
  geom[(size_t)i * 3] = gi.g0; geom[(size_t)i * 3 + 1] = gi.g1; geom[(size_t)i * 3 + 2] = gi.g2;
}

// =================================================================================================
// host side
// =================================================================================================
struct trx2_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // map
  int L = 0, use_orient = 0;
  std::string seq;
  float2 *Td = nullptr, *To = nullptr, *Tt = nullptr, *Tp = nullptr;
  float *pd = nullptr, *po = nullptr, *pt = nullptr, *pp = nullptr;
  unsigned char *gen = nullptr, *sel = nullptr, *mask2 = nullptr;
  float* knots_f = nullptr;
  double* knots_d = nullptr;
  double knots_h[TRX2_KTOT];
  // batch
  int Bcap = 0, Lcap = 0, BW = 64, Bpad = 0, nsplit = 1, nsplit_cap = 0;
  int* st_i = nullptr; double* st_d = nullptr; float* rho = nullptr;
  float4 *X = nullptr, *G = nullptr, *D = nullptr, *XT = nullptr, *S = nullptr, *Y = nullptr;
  float* xyz = nullptr; float4* xyzT = nullptr; float* wcur = nullptr; float4* geom = nullptr;
  float *fpart = nullptr, *epart = nullptr;
  double *e_last = nullptr, *f_last = nullptr;
  float* grad = nullptr; float* tors0 = nullptr;
  float4 *CX = nullptr, *CG = nullptr, *CD = nullptr, *CS = nullptr, *CY = nullptr;  // Cartesian runs (allocated on first use)
  int cart_B = 0, cart_L = 0;
  int* done_count = nullptr;
  int* seq_ctr = nullptr;
  // replayable graph of one chunk of (pair, step) launches; rebuilt when anything baked into the kernel arguments changes
  hipGraphExec_t gexec = nullptr;
  long g_key[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long alloc_epoch = 0;
  trx2_run* runs = nullptr;
  int* h_done = nullptr;  // pinned
  double last_seconds = 0; int last_launches = 0;
  // second lane (trx2_ctx_set_lanes): a context of its own stream and batch buffers that BORROWS this one's tables, so that
  // one job can run as two half-batches whose pair and step kernels overlap
  trx2_ctx* child = nullptr;
  bool borrows_map = false;
};

#define HIPCHK(expr)                                                                                        \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess) {                                                                                 \
      ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);                                         \
      return 1;                                                                                             \
    }                                                                                                       \
  } while (0)

static double round_txt(double v, int dp) {
  char buf[64];
  snprintf(buf, sizeof buf, dp == 3 ? "%.3f" : "%.5f", v);
  return strtod(buf, nullptr);
}

extern "C" int trx2_abi_version(void) { return 1; }

extern "C" int trx2_ctx_create(int device, trx2_ctx** out) {
  if (!out) return 1;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return 2;
  trx2_ctx* ctx = new trx2_ctx();
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return 3;
  }
  const double rama[TRX2_RAMA_NB][3] = TRX2_RAMA_INIT;
  float rm[TRX2_RAMA_NB * 3], rsc[TRX2_RAMA_NB * 4];
  for (int k = 0; k < TRX2_RAMA_NB; k++) {
    const double ph = rama[k][0] * M_PI / 180.0, ps = rama[k][1] * M_PI / 180.0;
    rm[k * 3] = (float)ph;
    rm[k * 3 + 1] = (float)ps;
    rm[k * 3 + 2] = (float)rama[k][2];
    rsc[k * 4] = (float)sin(ph); rsc[k * 4 + 1] = (float)cos(ph); rsc[k * 4 + 2] = (float)sin(ps); rsc[k * 4 + 3] = (float)cos(ps);
  }
  // the step kernels of short chains stage the L-BFGS history in 96 KB of dynamic LDS (above the 64 KB default limit)
  if (hipFuncSetAttribute((const void*)k_chain<1, CHAIN_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, HIST_LDS_BYTES(CHAIN_THREADS)) != hipSuccess ||
      hipFuncSetAttribute((const void*)k_step<1, CHAIN_THREADS, CHAIN_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, HIST_LDS_BYTES(CHAIN_THREADS)) != hipSuccess) {
    delete ctx;
    return 4;
  }
  float cb[4];
  {  // local frame of local_atoms(): CA at the origin, C on +x, N in the xy plane; CB = CA + ka (b x c) + kb b + kc c
    const double ang = TRX2_A_N_CA_C * M_PI / 180.0, N[3] = {TRX2_B_N_CA * cos(ang), TRX2_B_N_CA * sin(ang), 0}, C[3] = {TRX2_B_CA_C, 0, 0};
    const double b[3] = {-N[0], -N[1], -N[2]}, c[3] = {C[0], C[1], C[2]};
    const double a[3] = {b[1] * c[2] - b[2] * c[1], b[2] * c[0] - b[0] * c[2], b[0] * c[1] - b[1] * c[0]};
    double CB[3];
    for (int i = 0; i < 3; i++) CB[i] = TRX2_CB_KA * a[i] + TRX2_CB_KB * b[i] + TRX2_CB_KC * c[i];
    auto dot3d = [](const double* u, const double* v) { return u[0] * v[0] + u[1] * v[1] + u[2] * v[2]; };
    auto angle = [&](const double* p, const double* q) { return acos(dot3d(p, q) / sqrt(dot3d(p, p) * dot3d(q, q))); };
    const double d = sqrt(dot3d(CB, CB));
    // improper N-C-CA-CB (IUPAC sign, as dihedral_grad): F = N - C, G = C - CA (CA is the origin), H = CB - CA
    const double F[3] = {N[0] - C[0], N[1] - C[1], N[2] - C[2]}, G[3] = {C[0], C[1], C[2]}, H[3] = {CB[0], CB[1], CB[2]};
    auto cross3 = [](const double* u, const double* v, double* o) { o[0] = u[1] * v[2] - u[2] * v[1]; o[1] = u[2] * v[0] - u[0] * v[2]; o[2] = u[0] * v[1] - u[1] * v[0]; };
    double A_[3], B_[3], BA[3];
    cross3(F, G, A_); cross3(H, G, B_); cross3(B_, A_, BA);
    const double tor = atan2(dot3d(BA, G) / sqrt(dot3d(G, G)), dot3d(A_, B_));
    cb[0] = (float)d; cb[1] = (float)angle(N, CB); cb[2] = (float)angle(C, CB); cb[3] = (float)tor;
  }
  if (hipMemcpyToSymbol(HIP_SYMBOL(c_cb_ideal), cb, sizeof cb) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(c_rama), rm, sizeof rm) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(c_rama_sc), rsc, sizeof rsc) != hipSuccess ||
      hipHostMalloc((void**)&ctx->h_done, sizeof(int)) != hipSuccess) {
    delete ctx;
    return 4;
  }
  *out = ctx;
  return 0;
}

static void lend_map(trx2_ctx* c);
static void free_map(trx2_ctx* c) {
  if (c->child) {  // the borrower must be idle and forget the tables before they go
    if (c->child->stream) (void)hipStreamSynchronize(c->child->stream);
    trx2_ctx* k = c->child;
    k->Td = k->To = k->Tt = k->Tp = nullptr; k->pd = k->po = k->pt = k->pp = nullptr;
    k->gen = k->sel = k->mask2 = nullptr; k->knots_f = nullptr; k->knots_d = nullptr; k->L = 0; k->alloc_epoch++;
  }
  void* p[] = {c->Td, c->To, c->Tt, c->Tp, c->pd, c->po, c->pt, c->pp, c->gen, c->sel, c->mask2, c->knots_f, c->knots_d};
  for (void* q : p)
    if (q && !c->borrows_map) (void)hipFree(q);
  c->Td = c->To = c->Tt = c->Tp = nullptr;
  c->pd = c->po = c->pt = c->pp = nullptr;
  c->gen = c->sel = c->mask2 = nullptr;
  c->knots_f = nullptr; c->knots_d = nullptr;
  c->L = 0;
}
static void free_batch(trx2_ctx* c) {
  void* p[] = {c->st_i, c->st_d, c->rho, c->X, c->G, c->D, c->XT, c->S, c->Y, c->xyz, c->xyzT, c->geom, c->wcur, c->fpart,
               c->epart, c->e_last, c->f_last, c->grad, c->tors0, c->done_count, c->seq_ctr, c->runs};
  for (void* q : p)
    if (q) (void)hipFree(q);
  c->st_i = nullptr; c->st_d = nullptr; c->rho = nullptr;
  c->X = c->G = c->D = c->XT = c->S = c->Y = nullptr;
  c->xyz = nullptr; c->xyzT = nullptr; c->geom = nullptr; c->wcur = nullptr; c->fpart = c->epart = nullptr;
  c->e_last = c->f_last = nullptr; c->grad = nullptr; c->tors0 = nullptr; c->done_count = nullptr; c->seq_ctr = nullptr; c->runs = nullptr;
  c->Bcap = c->Lcap = 0;
  c->alloc_epoch++;
  void* q[] = {c->CX, c->CG, c->CD, c->CS, c->CY};
  for (void* v : q)
    if (v) (void)hipFree(v);
  c->CX = c->CG = c->CD = c->CS = c->CY = nullptr;
  c->cart_B = c->cart_L = 0;
}
static int ensure_cart(trx2_ctx* ctx, int B) {
  const int L = ctx->L;
  if (ctx->CX && B <= ctx->cart_B && L <= ctx->cart_L) return 0;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  void* q[] = {ctx->CX, ctx->CG, ctx->CD, ctx->CS, ctx->CY};
  for (void* v : q)
    if (v) (void)hipFree(v);
  ctx->CX = ctx->CG = ctx->CD = ctx->CS = ctx->CY = nullptr;
  const size_t n = (size_t)B * L * 4;
  HIPCHK(hipMalloc((void**)&ctx->CX, sizeof(float4) * n));
  HIPCHK(hipMalloc((void**)&ctx->CG, sizeof(float4) * n));
  HIPCHK(hipMalloc((void**)&ctx->CD, sizeof(float4) * n));
  HIPCHK(hipMalloc((void**)&ctx->CS, sizeof(float4) * n * LBM));
  HIPCHK(hipMalloc((void**)&ctx->CY, sizeof(float4) * n * LBM));
  HIPCHK(hipMemsetAsync(ctx->CX, 0, sizeof(float4) * n, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->CG, 0, sizeof(float4) * n, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->CD, 0, sizeof(float4) * n, ctx->stream));
  ctx->cart_B = B; ctx->cart_L = L;
  ctx->alloc_epoch++;
  return 0;
}

// the child sees the parent's tables (read-only on both streams)
static void lend_map(trx2_ctx* c) {
  trx2_ctx* k = c->child;
  if (!k) return;
  k->L = c->L; k->use_orient = c->use_orient; k->seq = c->seq;
  k->Td = c->Td; k->To = c->To; k->Tt = c->Tt; k->Tp = c->Tp; k->pd = c->pd; k->po = c->po; k->pt = c->pt; k->pp = c->pp;
  k->gen = c->gen; k->sel = c->sel; k->mask2 = c->mask2; k->knots_f = c->knots_f; k->knots_d = c->knots_d;
  memcpy(k->knots_h, c->knots_h, sizeof c->knots_h);
  k->alloc_epoch++;
}

extern "C" int trx2_ctx_set_lanes(trx2_ctx* ctx, int lanes) {
  if (!ctx) return 1;
  if (lanes != 1 && lanes != 2) { ctx->err = "trx2_ctx_set_lanes: 1 or 2"; return 1; }
  if (lanes == 2 && !ctx->child) {
    trx2_ctx* k = nullptr;
    if (trx2_ctx_create(ctx->device, &k) != 0) { ctx->err = "trx2_ctx_set_lanes: cannot create the second lane"; return 1; }
    k->borrows_map = true;
    ctx->child = k;
    if (ctx->L) { HIPCHK(hipStreamSynchronize(ctx->stream)); lend_map(ctx); }
  } else if (lanes == 1 && ctx->child) {
    trx2_ctx* k = ctx->child;
    ctx->child = nullptr;
    trx2_ctx_destroy(k);  // borrows_map: frees its batch buffers only
  }
  return 0;
}

extern "C" void trx2_ctx_destroy(trx2_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->child) { trx2_ctx* k = ctx->child; ctx->child = nullptr; trx2_ctx_destroy(k); }
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->gexec) (void)hipGraphExecDestroy(ctx->gexec);
  free_map(ctx);
  free_batch(ctx);
  if (ctx->h_done) (void)hipHostFree(ctx->h_done);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char* trx2_last_error(const trx2_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

static int set_map_impl(trx2_ctx* ctx, int L, const char* seq, const float* dist, const float* omega, const float* theta,
                        const float* phi, const trx2_params* prm, bool device_ptrs) {
  if (!ctx) return 1;
  if (L < 4 || L > 1024 || !dist || !prm) { ctx->err = "trx2_set_map: need 4 <= L <= 1024, dist and params"; return 1; }
  const bool orient = omega && theta && phi;
  if (!orient && (omega || theta || phi)) { ctx->err = "trx2_set_map: omega/theta/phi must be all given or all NULL"; return 1; }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  free_map(ctx);
  ctx->alloc_epoch++;
  const size_t LL = (size_t)L * L;
  ctx->L = L; ctx->use_orient = orient; ctx->seq = seq ? std::string(seq, strnlen(seq, L)) : std::string();
  // knot positions after the reference's "%.3f" / "%.5f" text round trip (utils_ros.py:70,92,111,135)
  double* kn = ctx->knots_h;
  for (int k = 0; k < 3; k++) kn[k] = round_txt(prm->drep[k], 3);
  for (int k = 0; k < 32; k++) kn[3 + k] = round_txt(4.25 + prm->dstep * k, 3);
  const double astep = prm->astep_deg * M_PI / 180.0;
  {
    double start = -M_PI - 1.5 * astep, stop = M_PI + 1.5 * astep, step = (stop - start) / (KO - 1);
    for (int k = 0; k < KO; k++) {
      double v = (k == KO - 1) ? stop : start + k * step;
      kn[KD + k] = round_txt(v, 5);
      kn[KD + KO + k] = round_txt(v, 3);
    }
    start = -1.5 * astep; stop = M_PI + 1.5 * astep; step = (stop - start) / (KP - 1);
    for (int k = 0; k < KP; k++) kn[KD + 2 * KO + k] = round_txt((k == KP - 1) ? stop : start + k * step, 3);
  }
  float knf[TRX2_KTOT];
  for (int k = 0; k < TRX2_KTOT; k++) knf[k] = (float)kn[k];
  HIPCHK(hipMalloc((void**)&ctx->knots_d, sizeof(double) * TRX2_KTOT));
  HIPCHK(hipMalloc((void**)&ctx->knots_f, sizeof(float) * TRX2_KTOT));
  HIPCHK(hipMemcpyAsync(ctx->knots_d, kn, sizeof(double) * TRX2_KTOT, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->knots_f, knf, sizeof knf, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMalloc((void**)&ctx->Td, LL * KD * sizeof(float2)));
  HIPCHK(hipMalloc((void**)&ctx->pd, LL * 4));
  HIPCHK(hipMalloc((void**)&ctx->gen, LL));
  HIPCHK(hipMalloc((void**)&ctx->sel, LL));
  HIPCHK(hipMalloc((void**)&ctx->mask2, LL));
  HIPCHK(hipMemsetAsync(ctx->Td, 0, LL * KD * sizeof(float2), ctx->stream));
  if (orient) {
    HIPCHK(hipMalloc((void**)&ctx->To, LL * KO * sizeof(float2)));
    HIPCHK(hipMalloc((void**)&ctx->Tt, LL * KO * sizeof(float2)));
    HIPCHK(hipMalloc((void**)&ctx->Tp, LL * KP * sizeof(float2)));
    HIPCHK(hipMalloc((void**)&ctx->po, LL * 4));
    HIPCHK(hipMalloc((void**)&ctx->pt, LL * 4));
    HIPCHK(hipMalloc((void**)&ctx->pp, LL * 4));
    HIPCHK(hipMemsetAsync(ctx->To, 0, LL * KO * sizeof(float2), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->Tt, 0, LL * KO * sizeof(float2), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->Tp, 0, LL * KP * sizeof(float2), ctx->stream));
  }
  const float* src[4] = {dist, omega, theta, phi};
  float* dev[4] = {nullptr, nullptr, nullptr, nullptr};
  const int nb[4] = {TRX2_ND_BINS, TRX2_NO_BINS, TRX2_NO_BINS, TRX2_NP_BINS};
  for (int c = 0; c < 4; c++) {
    if (!src[c]) continue;
    if (device_ptrs) dev[c] = const_cast<float*>(src[c]);
    else {
      HIPCHK(hipMalloc((void**)&dev[c], LL * nb[c] * 4));
      HIPCHK(hipMemcpyAsync(dev[c], src[c], LL * nb[c] * 4, hipMemcpyHostToDevice, ctx->stream));
    }
  }
  BuildArgs A;
  A.L = L; A.use_orient = orient;
  A.dist = dev[0]; A.omega = dev[1]; A.theta = dev[2]; A.phi = dev[3];
  A.ebase = prm->ebase; for (int k = 0; k < 3; k++) A.erep[k] = prm->erep[k];
  A.meff = prm->meff; A.pcut = prm->pcut;
  for (int k = 0; k < 32; k++) A.bkgr[k] = std::pow((4.25 + prm->dstep * k) / prm->dcut, prm->alpha);
  A.knots = ctx->knots_d;
  A.Td = ctx->Td; A.To = ctx->To; A.Tt = ctx->Tt; A.Tp = ctx->Tp;
  A.pd = ctx->pd; A.po = ctx->po; A.pt = ctx->pt; A.pp = ctx->pp; A.gen = ctx->gen; A.sel = ctx->sel;
  hipLaunchKernelGGL(k_build_tables, dim3((unsigned)((LL + 127) / 128)), dim3(128), 0, ctx->stream, A);
  hipLaunchKernelGGL(k_pack_masks, dim3((unsigned)((LL + 255) / 256)), dim3(256), 0, ctx->stream, L, ctx->sel, ctx->mask2);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (!device_ptrs)
    for (int c = 0; c < 4; c++)
      if (dev[c]) HIPCHK(hipFree(dev[c]));
  if (ctx->child) { HIPCHK(hipStreamSynchronize(ctx->stream)); lend_map(ctx); }  // tables complete before the other stream reads them
  return 0;
}

extern "C" int trx2_set_map(trx2_ctx* ctx, int L, const char* seq, const float* dist, const float* omega,
                            const float* theta, const float* phi, const trx2_params* prm) {
  return set_map_impl(ctx, L, seq, dist, omega, theta, phi, prm, false);
}
extern "C" int trx2_set_map_device(trx2_ctx* ctx, int L, const char* seq, const float* dist, const float* omega,
                                   const float* theta, const float* phi, const trx2_params* prm) {
  return set_map_impl(ctx, L, seq, dist, omega, theta, phi, prm, true);
}

extern "C" int trx2_get_tables(trx2_ctx* ctx, int channel, float* y_y2, float* knots, float* prob, unsigned char* gen,
                               unsigned char* sel) {
  if (!ctx || !ctx->L) return 1;
  if (channel < 0 || channel > 3 || (channel > 0 && !ctx->use_orient)) { ctx->err = "trx2_get_tables: bad channel"; return 1; }
  HIPCHK(hipSetDevice(ctx->device));
  const size_t LL = (size_t)ctx->L * ctx->L;
  const int K[4] = {KD, KO, KO, KP};
  const int off[4] = {0, KD, KD + KO, KD + 2 * KO};
  const float2* T[4] = {ctx->Td, ctx->To, ctx->Tt, ctx->Tp};
  const float* P[4] = {ctx->pd, ctx->po, ctx->pt, ctx->pp};
  if (y_y2) HIPCHK(hipMemcpy(y_y2, T[channel], LL * K[channel] * sizeof(float2), hipMemcpyDeviceToHost));
  if (knots) for (int k = 0; k < K[channel]; k++) knots[k] = (float)ctx->knots_h[off[channel] + k];
  if (prob) HIPCHK(hipMemcpy(prob, P[channel], LL * 4, hipMemcpyDeviceToHost));
  if (gen) HIPCHK(hipMemcpy(gen, ctx->gen, LL, hipMemcpyDeviceToHost));
  if (sel) HIPCHK(hipMemcpy(sel, ctx->sel, LL, hipMemcpyDeviceToHost));
  return 0;
}

static int pick_bw(int B) {
  int bw = 1;
  while (bw < B && bw < 64) bw <<= 1;
  return bw;
}

static int ensure_batch(trx2_ctx* ctx, int B) {
  const int L = ctx->L;
  const int BW = pick_bw(B);
  const int ngrp = (B + BW - 1) / BW;
  const int Bpad = ngrp * BW;
  // b-range splits: enough workgroups (>= ~2 per CU) while every wave keeps a few residues b
  int nsplit = 1;
  {
    // k_pair holds 2 workgroups per CU (220 VGPRs): 512 resident slots.  EMPIRICAL rule from profiles/README.md
    // (L=150, B=64, splits 2/3/4/6 timed on MI355X): with distances only, the largest split whose grid fits one
    // round is fastest (450 workgroups: 30.8 us vs 35.2 at 600); with the angle channels on, 600 smaller workgroups
    // win despite the partial second round (52.5 us vs 58.0).  Every wave keeps at least two residues b.
    const int PW = 64 / BW;
    const long slots = ctx->use_orient ? 640 : 512;
    for (int n = 1; n <= 16; n++)
      if ((long)L * n * ngrp <= slots && L / n >= PAIR_WAVES * PW * 2) nsplit = n;
    if (const char* e = getenv("TRX2_NSPLIT")) { int v = atoi(e); if (v >= 1 && v <= 16) nsplit = v; }  // A/B timing only
  }
  ctx->BW = BW; ctx->Bpad = Bpad; ctx->nsplit = nsplit;
  if (B <= ctx->Bcap && L <= ctx->Lcap && nsplit <= ctx->nsplit_cap) return 0;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  free_batch(ctx);
  const size_t BL = (size_t)B * L;
  HIPCHK(hipMalloc((void**)&ctx->st_i, sizeof(int) * B * SI_N));
  HIPCHK(hipMalloc((void**)&ctx->st_d, sizeof(double) * B * SD_N));
  HIPCHK(hipMalloc((void**)&ctx->rho, sizeof(float) * B * LBM));
  HIPCHK(hipMalloc((void**)&ctx->X, sizeof(float4) * BL));
  HIPCHK(hipMalloc((void**)&ctx->G, sizeof(float4) * BL));
  HIPCHK(hipMalloc((void**)&ctx->D, sizeof(float4) * BL));
  HIPCHK(hipMalloc((void**)&ctx->XT, sizeof(float4) * BL));
  HIPCHK(hipMalloc((void**)&ctx->S, sizeof(float4) * BL * LBM));
  HIPCHK(hipMalloc((void**)&ctx->Y, sizeof(float4) * BL * LBM));
  HIPCHK(hipMalloc((void**)&ctx->xyz, sizeof(float) * BL * 16));
  HIPCHK(hipMalloc((void**)&ctx->geom, sizeof(float4) * BL * 3));
  HIPCHK(hipMalloc((void**)&ctx->xyzT, sizeof(float4) * (size_t)Bpad * L * 4));
  HIPCHK(hipMalloc((void**)&ctx->wcur, sizeof(float) * Bpad * 8));
  HIPCHK(hipMalloc((void**)&ctx->fpart, sizeof(float) * (size_t)nsplit * Bpad * L * 16));
  HIPCHK(hipMalloc((void**)&ctx->epart, sizeof(float) * (size_t)nsplit * Bpad * L * 8));
  HIPCHK(hipMalloc((void**)&ctx->e_last, sizeof(double) * B * TRX2_NTERMS));
  HIPCHK(hipMalloc((void**)&ctx->f_last, sizeof(double) * B));
  HIPCHK(hipMalloc((void**)&ctx->grad, sizeof(float) * BL * 3));
  HIPCHK(hipMalloc((void**)&ctx->tors0, sizeof(float) * BL * 3));
  HIPCHK(hipMalloc((void**)&ctx->done_count, sizeof(int)));
  HIPCHK(hipMalloc((void**)&ctx->seq_ctr, sizeof(int)));
  HIPCHK(hipMemsetAsync(ctx->seq_ctr, 0, sizeof(int), ctx->stream));
  HIPCHK(hipMalloc((void**)&ctx->runs, sizeof(trx2_run) * TRX2_MAX_RUNS));
  HIPCHK(hipMemsetAsync(ctx->xyzT, 0, sizeof(float4) * (size_t)Bpad * L * 4, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->wcur, 0, sizeof(float) * Bpad * 8, ctx->stream));
  ctx->Bcap = B; ctx->Lcap = L; ctx->nsplit_cap = nsplit;
  ctx->alloc_epoch++;
  return 0;
}

static PairArgs pair_args(trx2_ctx* c, int B) {
  PairArgs P;
  P.L = c->L; P.B = B; P.nsplit = c->nsplit; P.Bpad = c->Bpad;
  P.xyzT = c->xyzT; P.Td = c->Td; P.To = c->To; P.Tt = c->Tt; P.Tp = c->Tp;
  P.seq_ctr = c->seq_ctr;
  P.mask = c->mask2; P.knots = c->knots_f; P.wcur = c->wcur; P.fpart = c->fpart; P.epart = c->epart;
  return P;
}
static ChainArgs chain_args(trx2_ctx* c, int B, int mode, int nruns, int max_evals) {
  ChainArgs A;
  A.seq_ctr = c->seq_ctr;
  A.L = c->L; A.B = B; A.Bpad = c->Bpad; A.BW = c->BW; A.nsplit = c->nsplit; A.mode = mode; A.nruns = nruns;
  A.max_evals = max_evals; A.runs = c->runs; A.st_i = c->st_i; A.st_d = c->st_d; A.rho = c->rho;
  A.X = c->X; A.G = c->G; A.D = c->D; A.XT = c->XT; A.S = c->S; A.Y = c->Y; A.xyz = c->xyz; A.geom = c->geom; A.xyzT = c->xyzT;
  A.wcur = c->wcur; A.fpart = c->fpart; A.epart = c->epart; A.e_last = c->e_last; A.f_last = c->f_last;
  A.grad_out = c->grad; A.done_count = c->done_count;
  return A;
}

template <int FAM>
static void launch_pair_fam(trx2_ctx* c, const PairArgs& P, dim3 grid, dim3 block, hipStream_t st) {
  switch (c->BW) {
    case 64: hipLaunchKernelGGL((k_pair<64, FAM>), grid, block, 0, st, P); break;
    case 32: hipLaunchKernelGGL((k_pair<32, FAM>), grid, block, 0, st, P); break;
    case 16: hipLaunchKernelGGL((k_pair<16, FAM>), grid, block, 0, st, P); break;
    case 8: hipLaunchKernelGGL((k_pair<8, FAM>), grid, block, 0, st, P); break;
    case 4: hipLaunchKernelGGL((k_pair<4, FAM>), grid, block, 0, st, P); break;
    case 2: hipLaunchKernelGGL((k_pair<2, FAM>), grid, block, 0, st, P); break;
    default: hipLaunchKernelGGL((k_pair<1, FAM>), grid, block, 0, st, P); break;
  }
}
static void launch_pair(trx2_ctx* c, int B) {
  PairArgs P = pair_args(c, B);
  dim3 grid(c->L, c->nsplit, c->Bpad / c->BW), block(PAIR_THREADS);
  launch_pair_fam<FAM_ALL>(c, P, grid, block, c->stream);
}
static CartArgs cart_args(trx2_ctx* c, int B, int nruns, int max_evals) {
  CartArgs A;
  A.L = c->L; A.B = B; A.Bpad = c->Bpad; A.BW = c->BW; A.nsplit = c->nsplit; A.nruns = nruns; A.max_evals = max_evals; A.seq_ctr = c->seq_ctr;
  A.runs = c->runs; A.st_i = c->st_i; A.st_d = c->st_d; A.rho = c->rho;
  A.CX = c->CX; A.CG = c->CG; A.CD = c->CD; A.CS = c->CS; A.CY = c->CY;
  A.xyz = c->xyz; A.xyzT = c->xyzT; A.X = c->X; A.XT = c->XT; A.geom = c->geom; A.wcur = c->wcur;
  A.fpart = c->fpart; A.epart = c->epart; A.e_last = c->e_last; A.f_last = c->f_last; A.done_count = c->done_count;
  return A;
}
static void launch_chain(trx2_ctx* c, int B, int mode, int nruns, int max_evals) {
  ChainArgs A = chain_args(c, B, mode, nruns, max_evals);
  const dim3 grid(B);
  const int L = c->L;
  if (L <= CHAIN_THREADS) hipLaunchKernelGGL((k_chain<1, CHAIN_THREADS>), grid, dim3(CHAIN_THREADS), HIST_LDS_BYTES(CHAIN_THREADS), c->stream, A);
  else if (L <= 2 * CHAIN_THREADS) hipLaunchKernelGGL((k_chain<2, CHAIN_THREADS>), grid, dim3(CHAIN_THREADS), 0, c->stream, A);
  else hipLaunchKernelGGL((k_chain<4, CHAIN_THREADS>), grid, dim3(CHAIN_THREADS), 0, c->stream, A);
}

static int upload_single_run(trx2_ctx* ctx, const float* w, int sep_lo, int sep_hi) {
  trx2_run r;
  memset(&r, 0, sizeof r);
  for (int k = 0; k < TRX2_NW; k++) r.w[k] = w[k];
  r.max_iter = 1; r.sep_lo = sep_lo; r.sep_hi = sep_hi;
  HIPCHK(hipMemcpyAsync(ctx->runs, &r, sizeof r, hipMemcpyHostToDevice, ctx->stream));
  return 0;
}

extern "C" int trx2_eval_batch(trx2_ctx* ctx, int B, const float* tors, const float* w, int sep_lo, int sep_hi,
                               double* e_terms, double* f_total, float* grad, float* xyz) {
  if (!ctx) return 1;
  if (!ctx->L) { ctx->err = "trx2_eval_batch: no map set"; return 1; }
  if (B < 1 || !tors || !w) { ctx->err = "trx2_eval_batch: bad arguments"; return 1; }
  HIPCHK(hipSetDevice(ctx->device));
  if (ensure_batch(ctx, B)) return 1;
  const int L = ctx->L;
  const size_t BL = (size_t)B * L;
  if (upload_single_run(ctx, w, sep_lo, sep_hi)) return 1;
  HIPCHK(hipMemcpyAsync(ctx->tors0, tors, sizeof(float) * BL * 3, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->st_i, 0, sizeof(int) * B * SI_N, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->st_d, 0, sizeof(double) * B * SD_N, ctx->stream));
  hipLaunchKernelGGL(k_init_torsions, dim3((unsigned)((BL + 255) / 256)), dim3(256), 0, ctx->stream, L, B, 0ull, 0u,
                     ctx->tors0, ctx->X, ctx->XT, ctx->geom);
  launch_chain(ctx, B, MODE_INIT, 1, 1 << 30);
  launch_pair(ctx, B);
  launch_chain(ctx, B, MODE_FINISH, 1, 1 << 30);
  HIPCHK(hipGetLastError());
  if (e_terms) HIPCHK(hipMemcpyAsync(e_terms, ctx->e_last, sizeof(double) * B * TRX2_NTERMS, hipMemcpyDeviceToHost, ctx->stream));
  if (f_total) HIPCHK(hipMemcpyAsync(f_total, ctx->f_last, sizeof(double) * B, hipMemcpyDeviceToHost, ctx->stream));
  if (grad) HIPCHK(hipMemcpyAsync(grad, ctx->grad, sizeof(float) * BL * 3, hipMemcpyDeviceToHost, ctx->stream));
  std::vector<float> tmp;
  if (xyz) {
    tmp.resize(BL * 16);
    HIPCHK(hipMemcpyAsync(tmp.data(), ctx->xyz, sizeof(float) * BL * 16, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (xyz)
    for (size_t i = 0; i < BL; i++) memcpy(xyz + i * 15, tmp.data() + i * 16, 15 * sizeof(float));
  return 0;
}

static int fold_impl(trx2_ctx* ctx, int B, const trx2_run* runs, int nruns, uint64_t seed, uint32_t decoy0,
                     const float* tors0, int max_evals, float* tors_out, float* xyz_out, double* e_terms,
                     double* f_final, int* status, int* n_evals, int* n_iters) {
  if (!ctx) return 1;
  if (!ctx->L) { ctx->err = "trx2_fold_batch: no map set"; return 1; }
  if (B < 1 || !runs || nruns < 1 || nruns > TRX2_MAX_RUNS) { ctx->err = "trx2_fold_batch: bad arguments"; return 1; }
  bool has_cart = false;
  for (int i = 0; i < nruns; i++) has_cart |= runs[i].cartesian != 0;
  if (has_cart && ctx->L > 2 * CHAIN_THREADS) { ctx->err = "trx2_fold_batch: Cartesian-space runs support chains of up to 512 residues"; return 1; }
  if (max_evals <= 0) max_evals = 1 << 30;
  HIPCHK(hipSetDevice(ctx->device));
  if (ensure_batch(ctx, B)) return 1;
  if (has_cart && ensure_cart(ctx, B)) return 1;
  const int L = ctx->L;
  const size_t BL = (size_t)B * L;
  auto t0 = std::chrono::steady_clock::now();
  HIPCHK(hipMemcpyAsync(ctx->runs, runs, sizeof(trx2_run) * nruns, hipMemcpyHostToDevice, ctx->stream));
  if (tors0) HIPCHK(hipMemcpyAsync(ctx->tors0, tors0, sizeof(float) * BL * 3, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->st_i, 0, sizeof(int) * B * SI_N, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->st_d, 0, sizeof(double) * B * SD_N, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->done_count, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(k_init_torsions, dim3((unsigned)((BL + 255) / 256)), dim3(256), 0, ctx->stream, L, B, seed, decoy0,
                     tors0 ? ctx->tors0 : (const float*)nullptr, ctx->X, ctx->XT, ctx->geom);
  launch_chain(ctx, B, MODE_INIT, nruns, max_evals);
  int launches = 0;
  const int chunk = 64;
  // hard cap on launches: every decoy stops by itself at max_evals; the extra margin covers skipped runs
  const long cap = (long)max_evals + 64;
  HIPCHK(hipMemsetAsync(ctx->seq_ctr, 0, sizeof(int), ctx->stream));
  auto enqueue_chunk = [&]() {
    for (int i = 0; i < chunk; i++) {
      launch_pair(ctx, B);  // bumps the device-side evaluation counter
      if (has_cart) {
        // fused launch: workgroups of 256 (512 for 256 < L <= 512) threads, one residue per thread in the Cartesian role
        const ChainArgs ca = chain_args(ctx, B, MODE_STEP, nruns, max_evals);
        const CartArgs cc = cart_args(ctx, B, nruns, max_evals);
        const dim3 g2(2 * B), b1(CHAIN_THREADS), b2(2 * CHAIN_THREADS);
        if (L <= CHAIN_THREADS) hipLaunchKernelGGL((k_step<1, CHAIN_THREADS, CHAIN_THREADS>), g2, b1, HIST_LDS_BYTES(CHAIN_THREADS), ctx->stream, ca, cc);
        else hipLaunchKernelGGL((k_step<2, CHAIN_THREADS, 2 * CHAIN_THREADS>), g2, b2, 0, ctx->stream, ca, cc);
      } else
        launch_chain(ctx, B, MODE_STEP, nruns, max_evals);
    }
  };
  // The chunk is a static graph (its only per-evaluation input, the sequence number, lives in device memory): capture it
  // once per (batch shape, protocol length, buffers) and replay it -- 128 launches become one hipGraphLaunch.  Measured on
  // MI355X it changes nothing (the loop is not launch-bound: profiles/README.md), so direct launches stay the default and
  // TRX2_GRAPH=1 opts in.
  static const bool no_graph = getenv("TRX2_GRAPH") == nullptr;
  if (!no_graph) {
    const long key[8] = {B, nruns, max_evals, has_cart ? 1 : 0, L, ctx->nsplit, ctx->BW, ctx->alloc_epoch};
    if (!ctx->gexec || memcmp(key, ctx->g_key, sizeof key) != 0) {
      if (ctx->gexec) { (void)hipGraphExecDestroy(ctx->gexec); ctx->gexec = nullptr; }
      hipGraph_t graph = nullptr;
      HIPCHK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
      enqueue_chunk();
      HIPCHK(hipStreamEndCapture(ctx->stream, &graph));
      HIPCHK(hipGraphInstantiate(&ctx->gexec, graph, nullptr, nullptr, 0));
      HIPCHK(hipGraphDestroy(graph));
      memcpy(ctx->g_key, key, sizeof key);
    }
  }
  while (true) {
    if (no_graph) enqueue_chunk();
    else HIPCHK(hipGraphLaunch(ctx->gexec, ctx->stream));
    launches += chunk;
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(ctx->h_done, ctx->done_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (*ctx->h_done >= B || launches >= cap) break;
  }
  // final report: energies of the accepted point X under the last run's weights
  HIPCHK(hipMemcpyAsync(ctx->XT, ctx->X, sizeof(float4) * BL, hipMemcpyDeviceToDevice, ctx->stream));
  {
    std::vector<int> sti((size_t)B * SI_N);
    HIPCHK(hipMemcpyAsync(sti.data(), ctx->st_i, sizeof(int) * B * SI_N, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < B; i++) {
      if (status) status[i] = sti[(size_t)i * SI_N + SI_PHASE] == PH_DONE ? sti[(size_t)i * SI_N + SI_STATUS] : TRX2_MAXEVAL;
      if (n_evals) n_evals[i] = sti[(size_t)i * SI_N + SI_NEVALS];
      if (n_iters) n_iters[i] = sti[(size_t)i * SI_N + SI_NITERS];
      sti[(size_t)i * SI_N + SI_RUN] = nruns - 1;
    }
    HIPCHK(hipMemcpyAsync(ctx->st_i, sti.data(), sizeof(int) * B * SI_N, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  launch_chain(ctx, B, MODE_INIT, nruns, max_evals);
  launch_pair(ctx, B);
  launch_chain(ctx, B, MODE_FINISH, nruns, max_evals);
  HIPCHK(hipGetLastError());
  std::vector<float> tmpx, tmpt;
  if (xyz_out) { tmpx.resize(BL * 16); HIPCHK(hipMemcpyAsync(tmpx.data(), ctx->xyz, sizeof(float) * BL * 16, hipMemcpyDeviceToHost, ctx->stream)); }
  if (tors_out) { tmpt.resize(BL * 4); HIPCHK(hipMemcpyAsync(tmpt.data(), ctx->X, sizeof(float4) * BL, hipMemcpyDeviceToHost, ctx->stream)); }
  if (e_terms) HIPCHK(hipMemcpyAsync(e_terms, ctx->e_last, sizeof(double) * B * TRX2_NTERMS, hipMemcpyDeviceToHost, ctx->stream));
  if (f_final) HIPCHK(hipMemcpyAsync(f_final, ctx->f_last, sizeof(double) * B, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (xyz_out) for (size_t i = 0; i < BL; i++) memcpy(xyz_out + i * 15, tmpx.data() + i * 16, 15 * sizeof(float));
  if (tors_out) for (size_t i = 0; i < BL; i++) memcpy(tors_out + i * 3, tmpt.data() + i * 4, 3 * sizeof(float));
  ctx->last_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  ctx->last_launches = launches;
  return 0;
}

// Two lanes: decoys [0, B0) on this context, [B0, B) on the child, the child driven from its own host thread.  A decoy is
// still identified by (seed, decoy0 + index), so which lane folds it does not change its start; but the half-batches have
// their own decoy-group width and slab split, so results are those of folding the halves separately, not of one batch of B.
#define TRX2_LANE_MIN_B 32
extern "C" int trx2_fold_batch(trx2_ctx* ctx, int B, const trx2_run* runs, int nruns, uint64_t seed, uint32_t decoy0,
                               const float* tors0, int max_evals, float* tors_out, float* xyz_out, double* e_terms,
                               double* f_final, int* status, int* n_evals, int* n_iters) {
  if (!ctx) return 1;
  trx2_ctx* k = ctx->child;
  if (!k || B < TRX2_LANE_MIN_B || !ctx->L)
    return fold_impl(ctx, B, runs, nruns, seed, decoy0, tors0, max_evals, tors_out, xyz_out, e_terms, f_final, status, n_evals, n_iters);
  const int B0 = (B + 1) / 2, B1 = B - B0;
  const size_t L = (size_t)ctx->L;
  auto t0 = std::chrono::steady_clock::now();
  int rc1 = 0;
  std::thread other([&]() {
    rc1 = fold_impl(k, B1, runs, nruns, seed, decoy0 + (uint32_t)B0, tors0 ? tors0 + (size_t)B0 * L * 3 : nullptr, max_evals,
                    tors_out ? tors_out + (size_t)B0 * L * 3 : nullptr, xyz_out ? xyz_out + (size_t)B0 * L * 15 : nullptr,
                    e_terms ? e_terms + (size_t)B0 * TRX2_NTERMS : nullptr, f_final ? f_final + B0 : nullptr,
                    status ? status + B0 : nullptr, n_evals ? n_evals + B0 : nullptr, n_iters ? n_iters + B0 : nullptr);
  });
  const int rc0 = fold_impl(ctx, B0, runs, nruns, seed, decoy0, tors0, max_evals, tors_out, xyz_out, e_terms, f_final, status, n_evals, n_iters);
  other.join();
  if (rc1 != 0 && rc0 == 0) ctx->err = "second lane: " + k->err;
  ctx->last_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  ctx->last_launches = ctx->last_launches > k->last_launches ? ctx->last_launches : k->last_launches;
  return rc0 != 0 ? rc0 : rc1;
}

extern "C" int trx2_time_pair_kernel(trx2_ctx* ctx, int B, const float* w, int sep_lo, int sep_hi, int n_rep,
                                     double* ms_avg, double* term_evals) {
  if (!ctx) return 1;
  if (!ctx->L || !ctx->xyzT || B > ctx->Bcap || B < 1 || n_rep < 1) { ctx->err = "trx2_time_pair_kernel: run an eval/fold batch of this size first"; return 1; }
  HIPCHK(hipSetDevice(ctx->device));
  const int L = ctx->L;
  // weights / separation window for every decoy
  std::vector<float> wc((size_t)ctx->Bpad * 8, 0.0f);
  for (int i = 0; i < B; i++) {
    float* p = wc.data() + (size_t)i * 8;
    p[0] = w[0]; p[1] = w[1]; p[2] = w[2]; p[3] = w[3]; p[4] = (float)sep_lo; p[5] = (float)sep_hi; p[6] = 1.0f;
  }
  HIPCHK(hipMemcpyAsync(ctx->wcur, wc.data(), wc.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  launch_pair(ctx, B);  // warm
  HIPCHK(hipEventRecord(e0, ctx->stream));
  for (int i = 0; i < n_rep; i++) launch_pair(ctx, B);
  HIPCHK(hipEventRecord(e1, ctx->stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  HIPCHK(hipEventDestroy(e0));
  HIPCHK(hipEventDestroy(e1));
  if (ms_avg) *ms_avg = (double)ms / n_rep;
  if (term_evals) {
    std::vector<unsigned char> sel((size_t)L * L);
    HIPCHK(hipMemcpy(sel.data(), ctx->sel, sel.size(), hipMemcpyDeviceToHost));
    double n = 0;
    for (int a = 0; a < L; a++)
      for (int b = 0; b < L; b++) {
        int sep = abs(a - b);
        if (sep < sep_lo || sep >= sep_hi) continue;
        unsigned m = sel[(size_t)a * L + b];
        n += (m & 1) + ((m >> 1) & 1) + ((m >> 2) & 1) + ((m >> 3) & 1);
      }
    *term_evals = n * B;
  }
  return 0;
}

extern "C" int trx2_last_fold_stats(trx2_ctx* ctx, double* seconds, int* n_launches) {
  if (!ctx) return 1;
  if (seconds) *seconds = ctx->last_seconds;
  if (n_launches) *n_launches = ctx->last_launches;
  return 0;
}

#ifdef TRX2_STAMP
extern "C" int trx2_debug_chain_stamps(unsigned long long* out32, int reset) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_cstamp), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_cstamp), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
extern "C" int trx2_debug_stamps(unsigned long long* out32) {
  return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : 1;
}
#endif
