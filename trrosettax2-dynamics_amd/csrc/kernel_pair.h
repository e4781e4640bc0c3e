// kernel_pair.h -- K3/K4: pair-term kernel, lane = decoy (restraint splines, soft-sphere repulsion, backbone hydrogen bonds)
// -- included by trx2fold.hip.
// Not a stand-alone header: it relies on the macros, constant tables and helpers defined above its #include.
#pragma once
// =================================================================================================
// K3/K4: pair terms.  Workgroup = (row residue a, b-range split, decoy group); lane = decoy (BW decoys per
// wave, 64/BW residues b per wave step).  Each ORDERED pair (a,b) is visited from a's row and only the
// gradient on a's atoms is kept -> no atomics, no cross-workgroup reduction, deterministic.
// =================================================================================================
#define PR_NCOMP 18 /* gradient components per residue: N CA C O CB H */
#define PR_REC 24   /* floats per record: 18 gradient + 6 energies (dist omega theta phi vdw hb) */

struct PairArgs {
  int L, B, nsplit, Bpad;
  int kd;       // knots of the distance spline: TRX2_KD, or TRX2_KD_AF2 for gen_rst_af2 tables
  int dist_ca;  // 1: the distance restraint acts on C-alpha (gen_rst_af2), 0: on C-beta
  const float4* xyzT;  // [ngrp][L][5][BW] float4 : residue record N CA C O CB (+pad) | H, hasH ; decoy-minor
  const float2 *Td, *To, *Tt, *Tp;
  const unsigned char* mask;  // [L][L] packed: low nibble = selected bits of (a,b), high nibble = those of (b,a)
  const unsigned char* mask_odr;  // the same without the pairs flagged disordered (mode 3, first stage), or NULL
  const float* knots;         // [kd + 72] float
  const float* wcur;          // [B][8] : w_ap w_dih w_ang w_vdw sep_lo sep_hi active (2: ordered pairs only) w_hb
  float* FA;                  // [nsplit][B][L][24] per (slab, decoy, residue a): gradient on N CA C O CB H, then the raw energies
                              // dist omega theta phi vdw hb (PR_REC; the step kernel sums the slabs: sum_pair_records)
  int* seq_ctr;               // evaluation counter in device memory: bumped here, read by the step kernel that follows
};

// ikn[i] = 1 / (kn[i+1] - kn[i]), precomputed once per workgroup
__device__ __forceinline__ void spline_eval_dev(const float2* __restrict__ row, const float* kn, const float* ikn, int K,
                                                int idx, float x, float& e, float& de) {
  // idx is a guess; fix up against the (rounded, slightly non-uniform) knots
  idx = max(0, min(K - 2, idx));
  if (x < kn[idx]) idx = max(0, idx - 1);
  else if (x >= kn[idx + 1]) idx = min(K - 2, idx + 1);
  float lo = kn[idx], hi = kn[idx + 1];
  float2 k0 = row[idx], k1 = row[idx + 1];
  // the segment's cubic in t = x - lo, formed from (y, y'') of its two knots and evaluated by Horner:
  //   c1 = (y1-y0)/h - h (2 y0'' + y1'')/6,  c2 = y0''/2,  c3 = (y1''-y0'')/(6h)
  float h = hi - lo, ih = ikn[idx], t = x - lo;
  bool inside = (x > kn[0]) && (x < kn[K - 1]);
  float c1 = fmaf(-h * (1.0f / 6.0f), fmaf(2.0f, k0.y, k1.y), (k1.x - k0.x) * ih);
  float c3 = (k1.y - k0.y) * (ih * (1.0f / 6.0f));
  float ev = fmaf(fmaf(fmaf(c3, t, 0.5f * k0.y), t, c1), t, k0.x);
  float dv = fmaf(fmaf(3.0f * c3, t, k0.y), t, c1);
  // outside the knot range: constant end value, zero slope (SplineFunc)
  e = inside ? ev : (x <= kn[0] ? k0.x : k1.x);
  de = inside ? dv : 0.0f;
}

// The same evaluation in three stages, so that the lookups of all the terms of a visit can travel together: seek (ONE round of
// LDS reads: the guessed segment's ends and its neighbours' far ends), fetch (the segment's two knots: one 16-byte gather) and
// value (arithmetic only).  Term by term, a visit with all channels on was six dependent LDS -> LDS -> gather chains in a row.
struct SplSeg { int idx; float t, h, lo0, hi0; };
__device__ __forceinline__ SplSeg spline_seek(const float* kn, int K, int guess, float x) {
  const int idx = max(0, min(K - 2, guess));
  const float k0 = kn[idx], k1 = kn[idx + 1], km = kn[max(idx - 1, 0)], kp = kn[min(idx + 2, K - 1)];
  const bool dn = x < k0, up = !dn && x >= k1;
  SplSeg g;
  g.idx = dn ? max(0, idx - 1) : (up ? min(K - 2, idx + 1) : idx);
  const float lo = dn ? km : (up && idx < K - 2 ? k1 : k0);
  const float hi = dn ? (idx > 0 ? k0 : k1) : (up ? kp : k1);
  g.t = x - lo; g.h = hi - lo;
  g.lo0 = kn[0]; g.hi0 = kn[K - 1];
  return g;
}
__device__ __forceinline__ void spline_value(const SplSeg& g, float2 k0, float2 k1, float x, float& e, float& de) {
  const float h = g.h, ih = frcp(h), t = g.t;
  const bool inside = (x > g.lo0) && (x < g.hi0);
  const float c1 = fmaf(-h * (1.0f / 6.0f), fmaf(2.0f, k0.y, k1.y), (k1.x - k0.x) * ih);
  const float c3 = (k1.y - k0.y) * (ih * (1.0f / 6.0f));
  const float ev = fmaf(fmaf(fmaf(c3, t, 0.5f * k0.y), t, c1), t, k0.x);
  const float dv = fmaf(fmaf(3.0f * c3, t, k0.y), t, c1);
  e = inside ? ev : (x <= g.lo0 ? k0.x : k1.x);
  de = inside ? dv : 0.0f;
}

// one donor -> acceptor candidate of a backbone hydrogen bond (trx2_model.h TRX2_HB_*; oracle: orc_hbond_term): N-H of one
// residue, O=C of the other.  Returns the raw energy (<= 0) and ADDS its gradient scaled by s to gN, gH, gO, gC.
__device__ __forceinline__ float hbond_dev(f3 N, f3 H, f3 O, f3 C, float s, f3& gN, f3& gH, f3& gO, f3& gC) {
  const f3 u = H - N, v = O - H, w = O - C;
  const float d2 = dot(v, v), id = frsq(d2), d = d2 * id, x = (d - (float)TRX2_HB_D0) * (1.0f / (float)TRX2_HB_R);
  if (!(x > -1.0f && x < 1.0f)) return 0.0f;
  const float ilu = frsq(dot(u, u)), ilw = frsq(dot(w, w));
  const f3 uh = u * ilu, vh = v * id, wh = w * ilw;
  const float ct = dot(uh, vh), cp = -dot(wh, vh);
  if (!(ct > 0.0f && cp > 0.0f)) return 0.0f;
  const float q = 1.0f - x * x, fd = q * q, dfd = -4.0f * q * x * (1.0f / (float)TRX2_HB_R);
  const float S = (float)TRX2_HB_SCALE, ct2 = ct * ct, cp2 = cp * cp;
  const float kd = -S * dfd * ct2 * cp2 * s, kt = -S * fd * 2.0f * ct * cp2 * s, kp = -S * fd * ct2 * 2.0f * cp * s;
  const f3 tu = (vh - uh * ct) * ilu, tv = (uh - vh * ct) * id, pw = (vh + wh * cp) * (-ilw), pv = (wh + vh * cp) * (-id);
  const f3 gv = vh * kd + tv * kt + pv * kp, gu = tu * kt, gw = pw * kp;
  gO += gv + gw; gH += gu - gv; gN += gu * -1.0f; gC += gw * -1.0f;
  return -S * fd * ct2 * cp2;
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
  const int decc = min(dec, A.B - 1);

  STAMP_DECL
  __shared__ float s_kn[TRX2_KTOT_MAX], s_ikn[TRX2_KTOT_MAX];
  __shared__ float s_red[PAIR_WAVES * 64 * RED_STRIDE];  // [wave][decoy][24 (+1 pad: bank-conflict-free)]
  __shared__ unsigned char s_mask[1024];  // packed masks of this workgroup's residues b (chunk <= L <= 1024)
  __shared__ unsigned char s_mask_o[1024];  // ... without the disordered pairs (only when the map has an idr mask)
  // One evaluation = one sequence number.  Kept in device memory (not a kernel argument) so that a chunk of
  // (pair, step) launches is a STATIC graph that can be replayed.  The step kernel of this evaluation starts after this
  // kernel has finished (same stream), so every one of its workgroups reads the same, final value.
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && A.seq_ctr) *A.seq_ctr += 1;
  // this lane's weights and residue a: requested first, so that their latency (they were written by the step kernel on
  // other CUs a moment ago) runs under the LDS fill and its barrier instead of after it
  const float4* wp = reinterpret_cast<const float4*>(A.wcur + (size_t)decc * 8);
  const float4 w0 = wp[0], w1 = wp[1];
  const float4* xa = A.xyzT + ((size_t)(grp * L + a) * 5) * BW + d;
  const float4 q0 = xa[0], q1 = xa[BW], q2 = xa[2 * BW], q3 = xa[3 * BW], q4 = xa[4 * BW];
  const int kd = A.kd, ktot = kd + 2 * KO + KP;
  for (int i = threadIdx.x; i < ktot; i += PAIR_THREADS) {
    s_kn[i] = A.knots[i];
    s_ikn[i] = i + 1 < ktot ? 1.0f / (A.knots[i + 1] - A.knots[i]) : 0.0f;  // entries straddling two tables are never read
  }
  {
    const int chunk0 = (L + A.nsplit - 1) / A.nsplit, lo0 = split * chunk0, hi0 = min(L, lo0 + chunk0);
    for (int i = lo0 + threadIdx.x; i < hi0; i += PAIR_THREADS) {
      s_mask[i - lo0] = A.mask[(size_t)a * L + i];
      if (A.mask_odr) s_mask_o[i - lo0] = A.mask_odr[(size_t)a * L + i];
    }
  }
  __syncthreads();
  const float* knd = s_kn;
  const float* kno = s_kn + kd;
  const float* knt = s_kn + kd + KO;
  const float* knp = s_kn + kd + 2 * KO;
  const float *iknd = s_ikn, *ikno = s_ikn + kd, *iknt = s_ikn + kd + KO, *iknp = s_ikn + kd + 2 * KO;
  const float inv_o = 1.0f / (kno[1] - kno[0]), inv_p = 1.0f / (knp[1] - knp[0]);
  // distance knots: three unevenly spaced repulsive ones, then a uniform grid (0 / 2 / 3.5 / 4.25 + 0.5 k; AF2: 0 / 2.325 / 3.575 / 3.875 + 0.3125 k)
  const float kd1 = knd[1], kd2 = knd[2], kd3 = knd[3], inv_d = 1.0f / (knd[4] - knd[3]);

  const float w_ap = w0.x, w_dih = w0.y, w_ang = w0.z, w_vdw = w0.w, w_hb = w1.w;
  const int sep_lo = (int)w1.x, sep_hi = (int)w1.y;
  const bool active = live && w1.z != 0.0f;
  const bool odr_only = w1.z == 2.0f && A.mask_odr != nullptr;

  // residue a
  const f3 Na = mk3(q0.x, q0.y, q0.z), CAa = mk3(q0.w, q1.x, q1.y), Ca = mk3(q1.z, q1.w, q2.x),
           Oa = mk3(q2.y, q2.z, q2.w), CBa = mk3(q3.x, q3.y, q3.z), Ha = mk3(q4.x, q4.y, q4.z);
  const bool donor_a = q4.w != 0.0f;

  f3 gN = mk3(0, 0, 0), gCA = gN, gC = gN, gO = gN, gCB = gN, gH = gN;
  float e_d = 0, e_o = 0, e_t = 0, e_p = 0, e_v = 0, e_h = 0;

  const int chunk = (L + A.nsplit - 1) / A.nsplit;
  const int b_lo = split * chunk, b_hi = min(L, b_lo + chunk);
  const unsigned aL = (unsigned)a * (unsigned)L;
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
    // residue b's N, CA, CB: requested before the masks are looked at (the address needs only b), so that the two reads travel
    // together; a visit that turns out to have nothing to do drops them.  Index arithmetic in 24-bit multiplies (full rate;
    // L <= 1024, a few decoy groups) instead of the quarter-rate 64-bit multiply-adds of size_t indexing.
    const float4* xb = A.xyzT + (__umul24((unsigned)(grp * L + bc), 5u * BW) + (unsigned)d);
    float4 r0 = xb[0], r1 = xb[BW], r3 = xb[3 * BW];
    unsigned m_ab = 0, m_ba = 0;
    if (valid && sep >= sep_lo && sep < sep_hi) {
      const unsigned mm = odr_only ? s_mask_o[bc - b_lo] : s_mask[bc - b_lo];
      m_ab = mm & 15u;
      m_ba = mm >> 4;
    }
    if (!(FAM & FAM_SYM)) { m_ab &= ~(TRX2_M_DIST | TRX2_M_OMEGA); m_ba &= ~(TRX2_M_DIST | TRX2_M_OMEGA); }
    if (!(FAM & FAM_ASYM)) { m_ab &= ~(TRX2_M_THETA | TRX2_M_PHI); m_ba &= ~(TRX2_M_THETA | TRX2_M_PHI); }
    const unsigned msym = (a < bc) ? m_ab : m_ba;  // DIST / OMEGA bits live on the (min,max) row
    const bool dovdw = (FAM & FAM_VDW) && valid && sep >= TRX2_VDW_MINSEP && (w_vdw != 0.0f || w_hb != 0.0f);
    STAMP(1)  // masks (2 byte loads) + loop control
    if (!__any((int)(m_ab | m_ba | (unsigned)dovdw))) continue;

    const f3 Nb = mk3(r0.x, r0.y, r0.z), CAb = mk3(r0.w, r1.x, r1.y), CBb = mk3(r3.x, r3.y, r3.z);
    STAMP(2)  // coordinates of residue b (4 x 16 B per lane)
    const unsigned iab = aL + (unsigned)bc, iba = __umul24((unsigned)bc, (unsigned)L) + (unsigned)a;
    const unsigned isym = (a < bc) ? iab : iba;
    const bool first = a < bc;  // symmetric energies are counted from the lower row only

    // Geometry the terms of one pair share.  All five angular terms are built on u = CB_a - CB_b, va = CA_a - CB_a and
    // vb = CA_b - CB_b: omega's two plane normals are va x u and vb x u, theta(a,b)'s second normal IS va x u, theta(b,a)'s
    // is -(vb x u), the phi angles sit between va / vb and u.  Evaluated once per visit instead of once per term
    // (dihedral_grad / angle_grad, which stay for the step kernels): the all-channel visit is ~25 % shorter.
    const f3 u = CBa - CBb;
    const float u2 = dot(u, u), iu = frsq(u2), du = u2 * iu;
    // ---- distance: seek + fetch now, value after the angular lookups have been sent off too
    const bool on_d = (FAM & FAM_SYM) && (msym & TRX2_M_DIST);
    f3 ud = u;
    float idd = iu, dd = du;
    if (A.dist_ca) { ud = CAa - CAb; const float d2 = dot(ud, ud); idd = frsq(d2); dd = d2 * idd; }
    SplSeg sd;
    float2 kd0 = make_float2(0, 0), kd1_ = kd0;
    if (on_d) {
      sd = spline_seek(knd, kd, dd < kd1 ? 0 : (dd < kd2 ? 1 : (dd < kd3 ? 2 : 3 + (int)((dd - kd3) * inv_d))), dd);
      const float2* row = A.Td + __umul24(isym, (unsigned)kd) + sd.idx;
      kd0 = row[0]; kd1_ = row[1];
    }
    STAMP(3)  // dist: seek, fetch
    const unsigned m_om = (FAM & FAM_SYM) ? (msym & TRX2_M_OMEGA) : 0u;
    const unsigned m_tp_ab = (FAM & FAM_ASYM) ? (m_ab & (TRX2_M_THETA | TRX2_M_PHI)) : 0u, m_tp_ba = (FAM & FAM_ASYM) ? (m_ba & (TRX2_M_THETA | TRX2_M_PHI)) : 0u;
    if (m_om | m_tp_ab | m_tp_ba) {
      // every angle of the pair (those whose channel is off get weight zero: the selections of a pair's channels go together,
      // and straight-line code lets the five lookups share their round trips)
      const f3 va = CAa - CBa, vb = CAb - CBb;
      const float va2 = dot(va, va), vb2 = dot(vb, vb), iva = frsq(va2), ivb = frsq(vb2), vau = dot(va, u), vbu = dot(vb, u);
      const f3 Aa = cross(va, u), Bb = cross(vb, u);
      const float iAa2 = frcp(fmaxf(dot(Aa, Aa), 1e-12f)), iBb2 = frcp(fmaxf(dot(Bb, Bb), 1e-12f));
      const f3 na = Na - CAa, Ta = cross(na, va), nb = Nb - CAb, Tb = cross(nb, vb);
      const float iT2 = frcp(fmaxf(dot(Ta, Ta), 1e-12f));
      const float x_o = fast_atan2f(dot(cross(Bb, Aa), u) * iu, dot(Aa, Bb));                  // CA_a - CB_a - CB_b - CA_b
      const float x_t1 = fast_atan2f(dot(cross(Aa, Ta), va) * iva, dot(Ta, Aa));               // N_a - CA_a - CB_a - CB_b
      const float x_t2 = fast_atan2f(-dot(cross(Bb, Tb), vb) * ivb, -dot(Tb, Bb));             // N_b - CA_b - CB_b - CB_a
      const float c_p1 = fminf(1.0f, fmaxf(-1.0f, -vau * iva * iu)), s_p1 = fsqrt(1.0f - c_p1 * c_p1), x_p1 = fast_atan2f(s_p1, c_p1);  // CA_a - CB_a - CB_b
      const float c_p2 = fminf(1.0f, fmaxf(-1.0f, vbu * ivb * iu)), s_p2 = fsqrt(1.0f - c_p2 * c_p2), x_p2 = fast_atan2f(s_p2, c_p2);   // CA_b - CB_b - CB_a
      const SplSeg g_o = spline_seek(kno, KO, (int)((x_o - kno[0]) * inv_o), x_o), g_t1 = spline_seek(knt, KO, (int)((x_t1 - knt[0]) * inv_o), x_t1),
                   g_t2 = spline_seek(knt, KO, (int)((x_t2 - knt[0]) * inv_o), x_t2), g_p1 = spline_seek(knp, KP, (int)((x_p1 - knp[0]) * inv_p), x_p1),
                   g_p2 = spline_seek(knp, KP, (int)((x_p2 - knp[0]) * inv_p), x_p2);
      const float2* r_o = A.To + __umul24(isym, (unsigned)KO) + g_o.idx;
      const float2* r_t1 = A.Tt + __umul24(iab, (unsigned)KO) + g_t1.idx;
      const float2* r_t2 = A.Tt + __umul24(iba, (unsigned)KO) + g_t2.idx;
      const float2* r_p1 = A.Tp + __umul24(iab, (unsigned)KP) + g_p1.idx;
      const float2* r_p2 = A.Tp + __umul24(iba, (unsigned)KP) + g_p2.idx;
      const float2 o0 = r_o[0], o1 = r_o[1], t10 = r_t1[0], t11 = r_t1[1], t20 = r_t2[0], t21 = r_t2[1], p10 = r_p1[0], p11 = r_p1[1], p20 = r_p2[0], p21 = r_p2[1];
      STAMP(4)  // angles, five seeks, five fetches
      float ev, de;
      {  // omega: F = va, G = u, H = vb
        spline_value(g_o, o0, o1, x_o, ev, de);
        const float on = m_om ? 1.0f : 0.0f;
        if (first) e_o += on * ev;
        const float sc = on * w_dih * de, ga = du * iAa2, ca = vau * iAa2 * iu, cb = vbu * iBb2 * iu;
        gCA = fma3(Aa, -ga * sc, gCA);
        gCB = fma3(Aa, (ga + ca) * sc, fma3(Bb, -cb * sc, gCB));
      }
      {  // theta(a,b): F = na, G = va, H = -u; second normal = va x u
        spline_value(g_t1, t10, t11, x_t1, ev, de);
        const float on = (m_tp_ab & TRX2_M_THETA) ? 1.0f : 0.0f;
        e_t += on * ev;
        const float sc = on * w_dih * de, Gn = va2 * iva, ga = Gn * iT2, gb = Gn * iAa2, ca = dot(na, va) * iT2 * iva, cb = -vau * iAa2 * iva;
        gN = fma3(Ta, -ga * sc, gN);
        gCA = fma3(Ta, (ga + ca) * sc, fma3(Aa, -cb * sc, gCA));
        gCB = fma3(Aa, (cb - gb) * sc, fma3(Ta, -ca * sc, gCB));
      }
      {  // theta(b,a): F = nb, G = vb, H = u; second normal = -(vb x u); only CB_a's share
        spline_value(g_t2, t20, t21, x_t2, ev, de);
        const float on = (m_tp_ba & TRX2_M_THETA) ? 1.0f : 0.0f;
        gCB = fma3(Bb, -(vb2 * ivb) * iBb2 * (on * w_dih * de), gCB);
      }
      {  // phi(a,b): between va and -u
        spline_value(g_p1, p10, p11, x_p1, ev, de);
        const float on = (m_tp_ab & TRX2_M_PHI) ? 1.0f : 0.0f;
        e_p += on * ev;
        const f3 vh = va * iva, wh = u * -iu;
        const float is = -frcp(fmaxf(s_p1, 1e-8f)) * (on * w_ang * de);
        const f3 d1 = (wh - vh * c_p1) * (is * iva), d3 = (vh - wh * c_p1) * (is * iu);
        gCA += d1;
        gCB += (d1 + d3) * -1.0f;
      }
      {  // phi(b,a): between vb and u; only CB_a's share
        spline_value(g_p2, p20, p21, x_p2, ev, de);
        const float on = (m_tp_ba & TRX2_M_PHI) ? 1.0f : 0.0f;
        const f3 vh = vb * ivb, wh = u * iu;
        gCB = fma3(vh - wh * c_p2, -frcp(fmaxf(s_p2, 1e-8f)) * iu * (on * w_ang * de), gCB);
      }
    }
    if (on_d) {
      float ev, de;
      spline_value(sd, kd0, kd1_, dd, ev, de);
      if (first) e_d += ev;
      if (A.dist_ca) gCA = fma3(ud, w_ap * de * idd, gCA);
      else gCB = fma3(ud, w_ap * de * idd, gCB);
    }
    STAMP(8)  // phi(b,a)
    if ((FAM & FAM_VDW) && dovdw) {
      f3 dca = CAa - CAb;
      if (dot(dca, dca) < (float)TRX2_VDW_CUT2) vmask |= 1u << v;
    }
  }
  {
    // per-lane walk: every lane follows its own contact bits (max-over-lanes steps): repulsion over the 5 x 5 atom pairs and
    // the two hydrogen-bond candidates of the pair, gradient on residue a's atoms; symmetric energies counted from the lower row
    while (vmask) {  // per-lane trip count; lanes without further contacts idle
      const int v = __ffs((int)vmask) - 1;
      vmask &= vmask - 1;
      const int b = bb + v * VSTRIDE + h;
      const float4* xb = A.xyzT + (__umul24((unsigned)(grp * L + b), 5u * BW) + (unsigned)d);
      const float4 r0 = xb[0], r1 = xb[BW], r2 = xb[2 * BW], r3 = xb[3 * BW], r4 = xb[4 * BW];
      const f3 Nb = mk3(r0.x, r0.y, r0.z), Cb = mk3(r1.z, r1.w, r2.x), Ob = mk3(r2.y, r2.z, r2.w), Hb = mk3(r4.x, r4.y, r4.z);
      if (w_vdw != 0.0f) {
        const f3 pa[5] = {Na, CAa, Ca, Oa, CBa};
        const f3 pb[5] = {Nb, mk3(r0.w, r1.x, r1.y), Cb, Ob, mk3(r3.x, r3.y, r3.z)};
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
        if (a < b) e_v += (float)TRX2_VDW_SCALE * ev;
        gN = fma3(ga[0], s, gN);
        gCA = fma3(ga[1], s, gCA);
        gC = fma3(ga[2], s, gC);
        gO = fma3(ga[3], s, gO);
        gCB = fma3(ga[4], s, gCB);
      }
      if (w_hb != 0.0f && abs(a - b) >= TRX2_HB_MINSEP) {
        f3 dump = mk3(0, 0, 0);  // the other residue's share: its own row computes it
        if (donor_a) { const float e = hbond_dev(Na, Ha, Ob, Cb, w_hb, gN, gH, dump, dump); if (a < b) e_h += e; }
        if (r4.w != 0.0f) { const float e = hbond_dev(Nb, Hb, Oa, Ca, w_hb, dump, dump, gO, gC); if (a < b) e_h += e; }
      }
    }
  }
  STAMP(9)  // vdw
  }

  STAMP(10) // loop exit
  // ---- reduce over the PW residue sub-lanes (inside the wave) and over the waves (through LDS); write decoy-major records.
  // The sub-lanes of a decoy sit BW lanes apart: a butterfly over the lane distances BW, 2 BW, .. 32 leaves their sum in
  // every one of them.  (Summing all PAIR_WAVES * PW slots from LDS instead made 6 threads add 256 slots each when one
  // decoy is folded -- the case of every feedback iteration.)
  // LDS image [wave][decoy][21]: the h = 0 lanes write 20 values at stride 21 (no bank conflict); the readers are
  // (decoy, quad) pairs, 4 lanes per decoy, so every store instruction writes whole 64-B (gradient) / 32-B (energy) runs.
  {
    float vals[PR_REC] = {gN.x, gN.y, gN.z, gCA.x, gCA.y, gCA.z, gC.x, gC.y, gC.z, gO.x, gO.y, gO.z,
                           gCB.x, gCB.y, gCB.z, gH.x, gH.y, gH.z, e_d, e_o, e_t, e_p, e_v, e_h};
#pragma unroll
    for (int o = BW; o < 64; o <<= 1)
#pragma unroll
      for (int k = 0; k < PR_REC; k++) vals[k] += __shfl_xor(vals[k], o, 64);
    if (h == 0) {
      float* s = s_red + ((size_t)wave * BW + d) * RED_STRIDE;
#pragma unroll
      for (int k = 0; k < PR_REC; k++) s[k] = vals[k];
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 6 * BW; t += PAIR_THREADS) {  // 6 quads per decoy = one 24-float record
    const int dd = t / 6, q = t % 6;
    const int dc = grp * BW + dd;
    if (dc >= A.B) continue;
    float acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int sl = 0; sl < PAIR_WAVES; sl++) acc[i] += s_red[((size_t)sl * BW + dd) * RED_STRIDE + q * 4 + i];
    reinterpret_cast<float4*>(A.FA + (((size_t)split * A.B + dc) * L + a) * PR_REC)[q] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
  STAMP(11)  // epilogue: LDS image, barrier, column sums, stores
  STAMP_FLUSH
}
