// kernel_pair2.h -- K3/K4: pair-term kernel, every UNORDERED residue pair visited once -- included by trx2fold.hip.
// Not a stand-alone header: it relies on the macros, constant tables and helpers defined above its #include.
#pragma once
// =================================================================================================
// Restraint splines (CB-CB distance, omega, theta both ways, phi both ways), soft-sphere repulsion and backbone hydrogen
// bonds of a pair (a, b), a < b, are evaluated ONCE, with the gradient on both residues (round 1 visited every ordered pair
// from a's row and kept a's half: no reduction, but every term twice).
//
// Work decomposition.  Workgroup = (tile of the pair matrix, decoy group).  Tile = TA residues a x TB residues b; decoy group
// = DW decoys.  Lane = (decoy d, sub-lane h): a wave instruction works on PW = 64 / DW consecutive residues b for DW decoys,
// h fastest, so the PW sub-lanes of a decoy are one DPP row (DW = 4) or the whole wave (DW = 1).
//   * residue a lives in registers: wave w of the workgroup takes a = a0 + w, a0 + w + 4, ... (RA of them), and sweeps the
//     tile's TB = 4 PW residues b for each; a's gradient and the pair energies accumulate in registers over the sweep, are
//     summed over the sub-lanes (DPP) and written once: one record per (a, tile).
//   * residue b's gradient accumulates in LDS, [b][component][decoy], with ds_add_f32 (fire and forget).  The sweep is cut
//     into 4 phases; in phase k wave w works on b sub-range (w + k) mod 4, so at any time the 4 waves own disjoint parts of
//     the accumulator (a Latin square), with a workgroup barrier between phases: every address has one writer at a time and a
//     fixed order of additions -> bitwise reproducible, no global atomics.  Written once per (b, tile) at the end.
// The step kernel sums, per residue, the a-records of the tiles in its row and the b-records of the tiles in its column.
// Small decoy groups are deliberate: the tile is then large in pair space for the same work per workgroup, so a residue
// collects few records (perimeter / area), and the coordinates are read straight from the decoy-major buffer (PW
// consecutive residues of one decoy = PW x 80 contiguous bytes): no decoy-minor copy.
// =================================================================================================
#define P2_NW 4
#define P2_THREADS (64 * P2_NW)
#define P2_NCOMP 18 /* gradient components per residue: N CA C O CB H */
#define P2_AREC 24  /* floats per a-record: 18 gradient + 6 energies (dist omega theta phi vdw hb) */
#define P2_BREC 20  /* floats per b-record: 18 gradient + 2 pad */
#ifndef P2_MIN_WAVES
#define P2_MIN_WAVES 2 /* waves per SIMD the register allocation must admit */
#endif

struct Pair2Args {
  int L, B, RA, TA, nI, nJ, ntp, G;   // ntp = number of tiles incl. padding (multiple of 8); G decoy groups
  const float4* P;                    // [B][L][5] float4: N CA C O CB (15 floats + pad) | H.xyz, hasH
  const float2 *Td, *To, *Tt, *Tp;
  const unsigned char* mask;          // [L][L] packed: low nibble = selected bits of (a,b), high nibble = those of (b,a)
  const float* knots;                 // [107]
  const float* wcur;                  // [B][8] : w_ap w_dih w_ang w_vdw sep_lo sep_hi active w_hb
  float* FA;                          // [nJ][B][L][24]
  float* FB;                          // [nI][B][L][20]
  const short2* tiles;                // [ntp] (a-tile, b-tile); x < 0: padding
  int* seq_ctr;                       // evaluation counter in device memory: bumped here, read by the step kernel that follows
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

// sum over the PW sub-lanes of a decoy, total in every one of them
template <int PW>
__device__ __forceinline__ float sub_sum(float v) {
  if (PW == 64) return wave_sum_dpp(v);
  v += dpp_move<0xB1>(v);   // lane ^ 1
  v += dpp_move<0x4E>(v);   // lane ^ 2
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror
  return v;                 // PW == 16: one DPP row
}

__device__ __forceinline__ void lds_add(float* p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// one donor -> acceptor candidate of a backbone hydrogen bond (trx2_model.h TRX2_HB_*; oracle: orc_hbond_term): N-H of one
// residue, O=C of the other.  Returns the raw energy (<= 0) and ADDS its gradient scaled by s to gN, gH, gO, gC.
__device__ __forceinline__ float hbond_dev(f3 N, f3 H, f3 O, f3 C, float s, f3& gN, f3& gH, f3& gO, f3& gC) {
  const f3 u = H - N, v = O - H, w = O - C;
  const float d2 = dot(v, v), id = rsqrtf(d2), d = d2 * id, x = (d - (float)TRX2_HB_D0) * (1.0f / (float)TRX2_HB_R);
  if (!(x > -1.0f && x < 1.0f)) return 0.0f;
  const float ilu = rsqrtf(dot(u, u)), ilw = rsqrtf(dot(w, w));
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

template <int DW>
struct P2Cfg {
  static constexpr int PW = 64 / DW;
  static constexpr int TB = P2_NW * PW;  // residues b per tile
  // floats per residue of the LDS accumulator: 18 components x DW decoys + a pad that sends the sub-lanes h of the first
  // half-wave (and the DW-decoy pairs beside them) to different banks: stride mod 32 = number of decoys per half-wave
  static constexpr int BS = DW == 4 ? 98 : 19;
};

template <int DW>
__global__ __launch_bounds__(P2_THREADS, P2_MIN_WAVES) void k_pair2(Pair2Args A) {
  using C = P2Cfg<DW>;
  constexpr int PW = C::PW, TB = C::TB, BS = C::BS;
  static_assert(DW == 4 || DW == 1, "decoy groups of 1 (folds of one or two decoys; 4 was measured and lost to the lane = decoy kernel)");
  const int L = A.L;
  const int tile = blockIdx.x % A.ntp, grp = blockIdx.x / A.ntp;  // the groups of a tile are ntp (a multiple of 8) blocks apart:
  const short2 tl = A.tiles[tile];                                 // same XCD under round-robin placement, so they share its L2
  if (tl.x < 0) return;                                            // padding tile: the whole workgroup leaves before any barrier
  const int it = tl.x, jt = tl.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = lane / PW, h = lane % PW;
  const int dec = grp * DW + d;
  const bool live = dec < A.B;
  const int decc = min(dec, A.B - 1);

  __shared__ float s_kn[TRX2_KTOT], s_ikn[TRX2_KTOT];
  __shared__ float s_b[TB * BS];
  if (blockIdx.x == 0 && threadIdx.x == 0 && A.seq_ctr) *A.seq_ctr += 1;
  const float4* wp = reinterpret_cast<const float4*>(A.wcur + (size_t)decc * 8);
  const float4 w0 = wp[0], w1 = wp[1];
  for (int i = threadIdx.x; i < TRX2_KTOT; i += P2_THREADS) {
    s_kn[i] = A.knots[i];
    s_ikn[i] = i + 1 < TRX2_KTOT ? 1.0f / (A.knots[i + 1] - A.knots[i]) : 0.0f;  // entries straddling two tables are never read
  }
  for (int i = threadIdx.x; i < TB * BS; i += P2_THREADS) s_b[i] = 0.0f;
  __syncthreads();
  const float *knd = s_kn, *kno = s_kn + KD, *knt = s_kn + KD + KO, *knp = s_kn + KD + 2 * KO;
  const float *iknd = s_ikn, *ikno = s_ikn + KD, *iknt = s_ikn + KD + KO, *iknp = s_ikn + KD + 2 * KO;
  const float inv_o = 1.0f / (kno[1] - kno[0]), inv_p = 1.0f / (knp[1] - knp[0]);

  const float w_ap = w0.x, w_dih = w0.y, w_ang = w0.z, w_vdw = w0.w, w_hb = w1.w;
  const int sep_lo = (int)w1.x, sep_hi = (int)w1.y;
  const bool active = live && w1.z != 0.0f;
  const bool contacts_on = w_vdw != 0.0f || w_hb != 0.0f;
  const int a0 = it * A.TA, b0 = jt * TB;

  for (int ia = 0; ia < A.RA; ia++) {
    const int a = a0 + ia * P2_NW + wave;
    const bool a_ok = a < L;
    const float4* pa = A.P + ((size_t)decc * L + min(a, L - 1)) * 5;
    // N, CA, CB of residue a stay in registers for the sweep; C, O, H are needed only by pairs in contact and are re-read there
    // (the kernel is register-bound: 256 VGPRs = 2 waves per SIMD)
    const float4 q0 = pa[0], q1 = pa[1], q3 = pa[3];
    const f3 Na = mk3(q0.x, q0.y, q0.z), CAa = mk3(q0.w, q1.x, q1.y), CBa = mk3(q3.x, q3.y, q3.z);
    f3 gN = mk3(0, 0, 0), gCA = gN, gC = gN, gO = gN, gCB = gN, gH = gN;
    float e_d = 0, e_o = 0, e_t = 0, e_p = 0, e_v = 0, e_h = 0;

#pragma unroll 1
    for (int k = 0; k < P2_NW; k++) {
      const int sr = (wave + k) & (P2_NW - 1);
      const int bl = sr * PW + h, b = b0 + bl;
      const bool pv = active && a_ok && b < L && b > a;
      const int sep = b - a;
      unsigned mm = 0;
      if (pv && sep >= sep_lo && sep < sep_hi) mm = A.mask[(size_t)a * L + b];
      const unsigned m_ab = mm & 15u, m_ba = mm >> 4;
      const bool near = pv && sep >= TRX2_VDW_MINSEP && contacts_on;
      if (__any((int)(mm | (unsigned)near))) {
        const int bc = min(b, L - 1);
        const float4* pb = A.P + ((size_t)decc * L + bc) * 5;
        const float4 r0 = pb[0], r1 = pb[1], r3 = pb[3];
        const f3 Nb = mk3(r0.x, r0.y, r0.z), CAb = mk3(r0.w, r1.x, r1.y), CBb = mk3(r3.x, r3.y, r3.z);
        const size_t iab = (size_t)a * L + bc, iba = (size_t)bc * L + min(a, L - 1);
        float* sb = s_b + bl * BS + d;         // this lane's slot of the LDS accumulator: components DW floats apart
        f3 bN = mk3(0, 0, 0), bCA = bN, bCB = bN;  // restraint gradient on residue b of this visit

        if (m_ab & TRX2_M_DIST) {
          const f3 u = CBa - CBb;
          const float d2 = dot(u, u), id = rsqrtf(d2), dd = d2 * id;
          const int idx = dd < 2.0f ? 0 : (dd < 3.5f ? 1 : (dd < 4.25f ? 2 : 3 + (int)((dd - 4.25f) * 2.0f)));
          float ev, de;
          spline_eval_dev(A.Td + iab * KD, knd, iknd, KD, idx, dd, ev, de);
          e_d += ev;
          const float s = w_ap * de * id;
          gCB = fma3(u, s, gCB); bCB = fma3(u, -s, bCB);
        }
        if (m_ab & TRX2_M_OMEGA) {
          f3 d1, d2, d3, d4;
          const float x = dihedral_grad(CAa, CBa, CBb, CAb, d1, d2, d3, d4);
          float ev, de;
          spline_eval_dev(A.To + iab * KO, kno, ikno, KO, (int)((x - kno[0]) * inv_o), x, ev, de);
          e_o += ev;
          const float s = w_dih * de;
          gCA = fma3(d1, s, gCA); gCB = fma3(d2, s, gCB); bCB = fma3(d3, s, bCB); bCA = fma3(d4, s, bCA);
        }
        if (m_ab & TRX2_M_THETA) {
          f3 d1, d2, d3, d4;
          const float x = dihedral_grad(Na, CAa, CBa, CBb, d1, d2, d3, d4);
          float ev, de;
          spline_eval_dev(A.Tt + iab * KO, knt, iknt, KO, (int)((x - knt[0]) * inv_o), x, ev, de);
          e_t += ev;
          const float s = w_dih * de;
          gN = fma3(d1, s, gN); gCA = fma3(d2, s, gCA); gCB = fma3(d3, s, gCB); bCB = fma3(d4, s, bCB);
        }
        if (m_ba & TRX2_M_THETA) {
          f3 d1, d2, d3, d4;
          const float x = dihedral_grad(Nb, CAb, CBb, CBa, d1, d2, d3, d4);
          float ev, de;
          spline_eval_dev(A.Tt + iba * KO, knt, iknt, KO, (int)((x - knt[0]) * inv_o), x, ev, de);
          e_t += ev;
          const float s = w_dih * de;
          bN = fma3(d1, s, bN); bCA = fma3(d2, s, bCA); bCB = fma3(d3, s, bCB); gCB = fma3(d4, s, gCB);
        }
        if (m_ab & TRX2_M_PHI) {
          f3 d1, d2, d3;
          const float x = angle_grad(CAa, CBa, CBb, d1, d2, d3);
          float ev, de;
          spline_eval_dev(A.Tp + iab * KP, knp, iknp, KP, (int)((x - knp[0]) * inv_p), x, ev, de);
          e_p += ev;
          const float s = w_ang * de;
          gCA = fma3(d1, s, gCA); gCB = fma3(d2, s, gCB); bCB = fma3(d3, s, bCB);
        }
        if (m_ba & TRX2_M_PHI) {
          f3 d1, d2, d3;
          const float x = angle_grad(CAb, CBb, CBa, d1, d2, d3);
          float ev, de;
          spline_eval_dev(A.Tp + iba * KP, knp, iknp, KP, (int)((x - knp[0]) * inv_p), x, ev, de);
          e_p += ev;
          const float s = w_ang * de;
          bCA = fma3(d1, s, bCA); bCB = fma3(d2, s, bCB); gCB = fma3(d3, s, gCB);
        }
        // ---- residue b's restraint share goes to the LDS accumulator; this wave is the only writer of these addresses in
        //      this phase.  (Flushed before the contact block so that its registers are free there.)
        if (__any((int)mm)) {
          const bool ang = ((m_ab & (TRX2_M_OMEGA | TRX2_M_THETA)) | (m_ba & (TRX2_M_THETA | TRX2_M_PHI))) != 0;
          if (mm) { lds_add(sb + 12 * DW, bCB.x); lds_add(sb + 13 * DW, bCB.y); lds_add(sb + 14 * DW, bCB.z); }
          if (__any((int)ang)) {
            if (mm) {
              lds_add(sb + 0 * DW, bN.x); lds_add(sb + 1 * DW, bN.y); lds_add(sb + 2 * DW, bN.z);
              lds_add(sb + 3 * DW, bCA.x); lds_add(sb + 4 * DW, bCA.y); lds_add(sb + 5 * DW, bCA.z);
            }
          }
        }
        // ---- residue pairs in contact (which they are depends on the decoy): soft-sphere repulsion over the 5 x 5 atom pairs
        //      and the two hydrogen-bond candidates.  |H-O| < 3.0 A or any overlap implies |CA-CA| < 8.5 A (trx2_model.h).
        bool contact = false;
        if (near) { const f3 dca = CAa - CAb; contact = dot(dca, dca) < (float)TRX2_VDW_CUT2; }
        if (contact) {
          const float4 q1c = pa[1], q2 = pa[2], q4 = pa[4], r1c = pb[1], r2 = pb[2], r4 = pb[4];
          const f3 Ca = mk3(q1c.z, q1c.w, q2.x), Oa = mk3(q2.y, q2.z, q2.w), Ha = mk3(q4.x, q4.y, q4.z);
          const f3 Cb = mk3(r1c.z, r1c.w, r2.x), Ob = mk3(r2.y, r2.z, r2.w), Hb = mk3(r4.x, r4.y, r4.z);
          f3 cb[6] = {mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0)};  // on b: N CA C O CB H
          if (w_vdw != 0.0f) {
            const f3 pA[5] = {Na, CAa, Ca, Oa, CBa};
            const f3 pB[5] = {Nb, CAb, Cb, Ob, CBb};
            f3 ga[5] = {mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0)};
            const float sw = w_vdw * (float)TRX2_VDW_SCALE;
            float ev = 0;
#pragma unroll
            for (int p = 0; p < 5; p++)
#pragma unroll
              for (int q = 0; q < 5; q++) {
                const f3 u = pA[p] - pB[q];
                constexpr VdwTab T = make_vdw_tab();
                const float r02 = T.r0sq[p * 5 + q], ir = T.ir0sq[p * 5 + q];
                const float c = fmaxf(r02 - dot(u, u), 0.0f);
                ev = fmaf(c * c, ir, ev);
                const float s = -4.0f * c * ir;
                ga[p] = fma3(u, s, ga[p]); cb[q] = fma3(u, -s, cb[q]);
              }
            e_v += (float)TRX2_VDW_SCALE * ev;
            gN = fma3(ga[0], sw, gN); gCA = fma3(ga[1], sw, gCA); gC = fma3(ga[2], sw, gC); gO = fma3(ga[3], sw, gO); gCB = fma3(ga[4], sw, gCB);
#pragma unroll
            for (int q = 0; q < 5; q++) cb[q] = cb[q] * sw;
          }
          if (w_hb != 0.0f && sep >= TRX2_HB_MINSEP) {
            if (q4.w != 0.0f) e_h += hbond_dev(Na, Ha, Ob, Cb, w_hb, gN, gH, cb[3], cb[2]);
            if (r4.w != 0.0f) e_h += hbond_dev(Nb, Hb, Oa, Ca, w_hb, cb[0], cb[5], gO, gC);
          }
          // component order of the accumulator: N CA C O CB H
          lds_add(sb + 0 * DW, cb[0].x); lds_add(sb + 1 * DW, cb[0].y); lds_add(sb + 2 * DW, cb[0].z);
          lds_add(sb + 3 * DW, cb[1].x); lds_add(sb + 4 * DW, cb[1].y); lds_add(sb + 5 * DW, cb[1].z);
          lds_add(sb + 6 * DW, cb[2].x); lds_add(sb + 7 * DW, cb[2].y); lds_add(sb + 8 * DW, cb[2].z);
          lds_add(sb + 9 * DW, cb[3].x); lds_add(sb + 10 * DW, cb[3].y); lds_add(sb + 11 * DW, cb[3].z);
          lds_add(sb + 12 * DW, cb[4].x); lds_add(sb + 13 * DW, cb[4].y); lds_add(sb + 14 * DW, cb[4].z);
          lds_add(sb + 15 * DW, cb[5].x); lds_add(sb + 16 * DW, cb[5].y); lds_add(sb + 17 * DW, cb[5].z);
        }
      }
      __syncthreads();  // the sub-ranges rotate: every wave's additions of this phase are done before another wave owns them
    }
    // ---- residue a of this sweep: sum over the sub-lanes, one record per (a, tile)
    float vals[P2_AREC] = {gN.x, gN.y, gN.z, gCA.x, gCA.y, gCA.z, gC.x, gC.y, gC.z, gO.x, gO.y, gO.z,
                           gCB.x, gCB.y, gCB.z, gH.x, gH.y, gH.z, e_d, e_o, e_t, e_p, e_v, e_h};
#pragma unroll
    for (int i = 0; i < P2_AREC; i++) vals[i] = sub_sum<PW>(vals[i]);
    if (h == 0 && live && a_ok) {
      float4* o = reinterpret_cast<float4*>(A.FA + (((size_t)jt * A.B + dec) * L + a) * P2_AREC);
#pragma unroll
      for (int q = 0; q < P2_AREC / 4; q++) o[q] = make_float4(vals[q * 4], vals[q * 4 + 1], vals[q * 4 + 2], vals[q * 4 + 3]);
    }
  }
  // ---- residues b of the tile: one record per (b, tile); consecutive records of a decoy are contiguous
  for (int i = threadIdx.x; i < DW * TB * (P2_BREC / 4); i += P2_THREADS) {
    const int q = i % (P2_BREC / 4), bl = (i / (P2_BREC / 4)) % TB, dd = i / ((P2_BREC / 4) * TB);
    const int b = b0 + bl, dc = grp * DW + dd;
    if (b >= L || dc >= A.B) continue;
    const float* sb = s_b + bl * BS + dd;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { const int c = q * 4 + j; v[j] = c < P2_NCOMP ? sb[c * DW] : 0.0f; }
    reinterpret_cast<float4*>(A.FB + (((size_t)it * A.B + dc) * L + b) * P2_BREC)[q] = make_float4(v[0], v[1], v[2], v[3]);
  }
}
