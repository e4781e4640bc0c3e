// kernel_pair.h -- K3/K4: pair-term kernel, lane = decoy (restraint splines, soft-sphere repulsion, backbone hydrogen bonds)
// -- included by trx2fold.hip.
// Not a stand-alone header: it relies on the macros, constant tables and helpers defined above its #include.
#pragma once
// =================================================================================================
// K3/K4: pair terms.  Workgroup = (row residue a, slice, decoy group); lane = decoy (BW decoys per wave, 64/BW partner
// residues b per wave step).  Each ORDERED pair (a,b) is handled from a's row and only the gradient on a's atoms is kept
// -> no atomics, no cross-workgroup reduction, deterministic.
// Round 3: the kernel no longer visits every pair.  (i) Restraint terms walk row a's COMPACTED LIST of partners that carry
// any selected restraint (k_build_rows, once per map / feedback step: 47 % of the ordered pairs at L=150 with distances
// only, ~25 % at L=400), the list cut into equal slices over the workgroups of a row.  (ii) Repulsion / hydrogen-bond
// CONTACTS depend on the decoy, so they cannot come from a list built per map; they are found by a scan of the partners'
// C-alpha alone -- the decoy-minor record starts with CA, one float4 -- with the loads of eight visits issued together
// (one memory round trip per eight visits instead of one per visit), then every lane walks its own contacts as before.
// What the reference asks Rosetta for with nb_list=True (folding/folding.py:91-102), without a list to keep valid.
// =================================================================================================
#define PR_NCOMP 18 /* gradient components per residue: N CA C O CB H */
#define PR_REC 24   /* floats per record: 18 gradient + 6 energies (dist omega theta phi vdw hb) */

struct PairArgs {
  int L, B, Bpad;
  int n_items;                // work items of this map's row plan (shared launches: the grid is the longest plan's, the rest exit)
  const uint2* items;         // work items of a launch, one workgroup each (row plan): x = row a | slice << 10 | slices of that row << 14,
                              // y = first entry of the slice in the row's list | one past its last << 16 (a row has < 1024 entries)
  int kd;       // knots of the distance spline: TRX2_KD, or TRX2_KD_AF2 for gen_rst_af2 tables
  int dist_ca;  // 1: the distance restraint acts on C-alpha (gen_rst_af2), 0: on C-beta
  const float4* xyzT;  // [ngrp][L][5][BW] float4, decoy-minor: residue record CA N CB C O (xt_pack) | H, hasH
  const float2 *Td, *To, *Tt, *Tp;
  const unsigned* rows;       // [L][L] row a: its row_cnt[a] partners with any selected restraint, ascending b:
                              //   b | mask << 16 | mask_odr << 24, mask = low nibble selected bits of (a,b), high nibble those of (b,a)
  const int* row_cnt;         // [L]
  const unsigned short* rows_rx;  // [L][L] beside rows: the entry's packed masks of the relax stage's re-selections, round 1 | round 2 << 8
  int has_odr;                // the map has an idr mask: entries carry the packed mask without the flagged pairs (mode 3, first stage)
  const float* knots;         // [kd + 72] float
  const float* wcur;          // [B][8] : w_ap w_dih w_ang w_vdw sep_lo sep_hi active (2: ordered pairs only) w_hb
  float* FA;                  // [slice][B][L][24] per (slice of the row, decoy, residue a): gradient on N CA C O CB H, then the raw energies
                              // dist omega theta phi vdw hb (PR_REC; the step kernel sums the slabs: sum_pair_records)
  int* seq_ctr;               // evaluation counter in device memory: bumped here, read by the step kernel that follows
  // Segment cache (round 4; NULL: off).  A spline lookup reads ONE cubic segment -- two knots, 16 bytes -- out of a 19-137 MB table at
  // an address that depends on the geometry: a dependent, uncoalesced gather that moves a 64-byte line for 16 bytes, six of them per
  // all-channel visit.  Between two evaluations of a minimisation a pair seldom leaves its segment, so every (row entry, decoy) keeps
  // the segments it used last: segc[a rowcap + e][NSEG] float4 (y0, y0'', y1, y1'') with their segment indices in
  // segt (one byte each).  The block's address depends on the entry alone, so it is requested WITH residue b's coordinates -- the
  // gather's round trip leaves the visit's dependent chain -- and a wave reads its lanes' blocks as one contiguous run.  A lookup whose
  // segment index differs from the tag gathers from the table as before and refreshes the block.  The cached numbers ARE the
  // table's: results are bit for bit those without the cache.  Valid as long as the tables are (the host resets the tags when they
  // change); nothing in a block depends on which decoy used it last.  Single-decoy folds only (pair_body, SEGC).
  float4* segc; uint2* segt; int rowcap;
};

// One spline evaluation in three stages, so that the lookups of all the terms of a visit can travel together: seek (ONE round of
// LDS reads: the guessed segment's ends and its neighbours' far ends), fetch (the segment's two knots: one 16-byte gather) and
// value (arithmetic only: the segment's cubic in t = x - lo from (y, y'') of its two knots by Horner; outside the knot range the
// constant end value and zero slope -- SplineFunc).  Term by term, a visit with all channels on was six dependent
// LDS -> LDS -> gather chains in a row.
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
// variants can be A/B-timed on hardware.  Round 3's kernel needs 169 VGPRs with all channels (round 2's: 217), one more than
// three waves per SIMD allow: at 3 it spills nothing at 64 and 1 decoys per wave and 1-9 registers in between, and the
// all-channel launch of 64 decoys takes 35.6 us against 43.2 at 2 (profiles/README.md, round 3).  The distance-only
// instantiation at 4 (128 registers) spills 31-69 and is 1.8 x slower; at 3 it spills nothing.
#ifndef PAIR_MIN_WAVES
#define PAIR_MIN_WAVES 3
#endif
#ifndef PAIR_MIN_WAVES_DIST
#define PAIR_MIN_WAVES_DIST 3
#endif
// Diagnostic build only (-DTRX2_STAMP, never the shipped library): wave 0 of the workgroup (a = L/2, split 0, group 0)
// accumulates s_memtime cycles per phase; every stamp first drains the memory counters so that a load's latency is charged
// to the phase that issued it.  The drains forbid overlaps the real kernel has: read SHARES, not the total.
#ifdef TRX2_STAMP
__device__ unsigned long long g_stamp[32];
#define STAMP_DECL unsigned long long st_acc[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}; unsigned long long st_prev = 0; \
  const bool st_on = (bx == gridDim.x / 2 && grp == 0 && (threadIdx.x >> 6) == 0); \
  if (st_on) { __builtin_amdgcn_s_waitcnt(0); st_prev = __builtin_amdgcn_s_memtime(); }
#define STAMP(k) if (st_on) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
  __builtin_amdgcn_s_waitcnt(0); st_acc[k] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); }
#define STAMP_FLUSH if (st_on && (threadIdx.x & 63) == 0) { for (int k_ = 0; k_ < 16; k_++) g_stamp[k_] = st_acc[k_]; }
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_FLUSH
#endif
// FAM selects the term families an instantiation evaluates.  Maps with the angle channels run FAM_ALL; distance-only maps
// (--no-orient, gen_rst_af2) run FAM_DIST | FAM_VDW, which leaves the whole angular block out of the instruction stream and
// out of the register allocation.
#define FAM_DIST 1  /* distance spline: needs CA (AF2 tables) or CB */
#define FAM_ANG 2   /* omega, theta, phi (both directions): needs N, CA, CB */
#define FAM_VDW 4   /* soft-sphere repulsion + backbone hydrogen bonds: all atoms, no tables */
#define FAM_ALL 7
#define PAIR_ROW_B_BITS 0x3ffu /* partner index in a row entry: L <= 1024 */

// Row lists (once per map and per feedback step, after k_pack_masks): row a keeps, in ascending order, every partner b != a
// whose packed mask byte -- or either relax-stage mask -- has any bit: a selected restraint of (a,b) or of (b,a).  One workgroup per row, order-preserving
// compaction by wave ballots.  Fixed row stride L: no prefix sum over rows.
__global__ __launch_bounds__(256) void k_build_rows(int L, const unsigned char* __restrict__ mask, const unsigned char* __restrict__ mask_odr,
                                                     const unsigned char* __restrict__ mask_r1, const unsigned char* __restrict__ mask_r2,
                                                     unsigned* __restrict__ rows, unsigned short* __restrict__ rows_rx, int* __restrict__ row_cnt) {
  const int a = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int s_w[4];
  __shared__ int s_base;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int b0 = 0; b0 < L; b0 += 256) {
    const int b = b0 + tid;
    unsigned m = 0, mo = 0, mr = 0;
    if (b < L && b != a) {
      m = mask[(size_t)a * L + b];
      mo = mask_odr ? mask_odr[(size_t)a * L + b] : m;
      mr = (unsigned)mask_r1[(size_t)a * L + b] | ((unsigned)mask_r2[(size_t)a * L + b] << 8);
    }
    // The relax re-selections are made from ALL generated restraints (add_rst(.., nogly=True), utils_ros.py:713-717), not from the
    // selection at the map's PCUT: with -pd above 0.15 they hold pairs that `m` does not.  Such an entry costs the
    // other stages nothing: the visit takes its mask from the run's filter.
    const bool keep = (m | mr) != 0;
    const unsigned long long bal = __ballot(keep);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_w[wave] = __popcll(bal);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; w++) off += s_w[w];
    if (keep) {
      rows[(size_t)a * L + off + pre] = (unsigned)b | (m << 16) | (mo << 24);
      rows_rx[(size_t)a * L + off + pre] = (unsigned short)mr;
    }
    __syncthreads();
    if (tid == 0) s_base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
  }
  if (tid == 0) row_cnt[a] = s_base;
}

// Register budget: three waves per SIMD where the instantiation fits 168 registers without spilling (all channels at 64 and at 1
// decoys per wave, every distance-only one), two otherwise: the all-channel kernels at 2 .. 32 decoys per wave spill 1-9
// registers at three, inside the visit loop -- 19 MB of scratch stores and as many reloads per launch at L=400 with 16 decoys
// per wave (WRITE_SIZE 19.5 MB against 0.8 MB of records, profiles/r03_traffic.json) for no gain in time (32.5 against 31.9 us).
template <int BW, int FAM>
constexpr int pair_min_waves() { return (FAM & FAM_ANG) ? ((BW == 64 || BW == 1) ? PAIR_MIN_WAVES : 2) : PAIR_MIN_WAVES_DIST; }
// The kernel's body: work item `bx` of the row plan, decoy group `grp` of the map and batch that A describes.
// one spline segment: from the cache block when its tag matches the segment the geometry asks for, else from the table (miss = true)
__device__ __forceinline__ void seg_fetch(bool use_c, unsigned tag, int idx, float4 c, const float2* row, float2& k0, float2& k1, bool& miss) {
  if (use_c && tag == (unsigned)idx) { k0 = make_float2(c.x, c.y); k1 = make_float2(c.z, c.w); }
  else { k0 = row[0]; k1 = row[1]; miss = use_c; }
}
// NW = waves of the workgroup.  4 (PAIR_WAVES) for batches: the waves share a row's list and partner range.  1 for SINGLE-DECOY
// folds (round 4): a lone decoy's row holds ~70 entries and its chain 90-400 partner residues, so four waves of 64 lanes each ran
// a quarter full; one wave per row walks the list in 2-3 steps, needs no workgroup barrier, no LDS image of the partial sums and no
// cross-wave reduction (the 24 sums leave the wave through DPP row rotations and v_readlane), and -- what matters when the
// single-decoy folds of many chains share a launch (launch_engine.h) -- a launch holds a quarter of the waves.
// SEGC: the instantiation that keeps a segment cache (PairArgs) -- single-decoy folds only (BW == 1).  With several decoys per wave
// the lanes of a sub-lane look up the SAME pair's table row, a few cache lines for the whole wave, and a block per (entry, decoy)
// would fetch sixteen times as much (measured: config 3 -26 % with it); with one decoy per wave every lane is another pair.
// SUBW > 1 (k_half_multi, kernel_step.h): the workgroup holds SUBW INDEPENDENT one-wave work items (NW == 1), one per wave, each with
// its own LDS arrays and no barrier between them; rows of at most PAIR_SUB_ENT entries (chains of up to 256 residues).
#define PAIR_SUB_ENT 256
template <int BW, int FAM, int NW = PAIR_WAVES, bool SEGC = false, int SUBW = 1>
__device__ __forceinline__ void pair_body(const PairArgs& A, const unsigned bx, const int grp) {
  static_assert(NW == PAIR_WAVES || (NW == 1 && BW == 1), "the one-wave workgroup serves single-decoy folds");
  static_assert(SUBW == 1 || NW == 1, "independent work items per wave: the one-wave form only");
  static_assert(!SEGC || BW == 1, "the segment cache serves single-decoy folds");
  constexpr int NT = NW * 64;   // threads of the workgroup
  constexpr int PW = 64 / BW;
  const int L = A.L;
  // Work item = (row a, slice, slices of that row): rows are cut into a number of slices that follows their list length (the
  // row plan, host side), so that no workgroup walks a list twice as long as the others' (rows hold 20 .. 140 partners)
  const uint2 item2 = A.items[bx];
  const unsigned item = item2.x;
  const int a = (int)(item & PAIR_ROW_B_BITS), split = (int)((item >> 10) & 15u), nsl = (int)(item >> 14);
  const int lane = threadIdx.x & 63, wave = SUBW > 1 ? 0 : (int)(threadIdx.x >> 6);
  const int sub = SUBW > 1 ? (int)(threadIdx.x >> 6) : 0;        // which of the workgroup's independent work items this wave is
  const int tix = SUBW > 1 ? lane : (int)threadIdx.x;           // thread index within the work item
  const int d = lane % BW, h = lane / BW;
  const int dec = grp * BW + d;
  const bool live = dec < A.B;
  const int decc = min(dec, A.B - 1);

  STAMP_DECL
  constexpr int ENT_CAP = SUBW > 1 ? PAIR_SUB_ENT : 1024;
  __shared__ float s_kn_[SUBW][TRX2_KTOT_MAX];
  __shared__ float s_red[NW == 1 ? 1 : NW * 64 * RED_STRIDE];  // [wave][decoy][24 (+1 pad: bank-conflict-free)]; unused by a one-wave workgroup
  __shared__ unsigned s_ent_[SUBW][ENT_CAP];             // this workgroup's slice of row a's list (a row has < L <= 1024 entries)
  __shared__ unsigned short s_rx_[SUBW][ENT_CAP];        // ... and the entries' relax-stage masks
  float* const s_kn = s_kn_[sub];
  unsigned* const s_ent = s_ent_[sub];
  unsigned short* const s_rx = s_rx_[sub];
  // One evaluation = one sequence number.  Kept in device memory (not a kernel argument) so that a chunk of
  // (pair, step) launches is a STATIC graph that can be replayed.  The step kernel of this evaluation starts after this
  // kernel has finished (same stream), so every one of its workgroups reads the same, final value.
  if (bx == 0 && grp == 0 && tix == 0 && A.seq_ctr) *A.seq_ctr += 1;
  // this lane's weights and residue a: requested first, so that their latency (they were written by the step kernel on
  // other CUs a moment ago) runs under the LDS fill and its barrier instead of after it
  const float4* wp = reinterpret_cast<const float4*>(A.wcur + (size_t)decc * 8);
  const float4 w0 = wp[0], w1 = wp[1];
  const float4* xa = A.xyzT + ((size_t)(grp * L + a) * 5) * BW + d;
  const float4 q0 = xa[0], q1 = xa[BW], q2 = xa[2 * BW], q3 = xa[3 * BW], q4 = xa[4 * BW];
  const int kd = A.kd, ktot = kd + 2 * KO + KP;
  // Every decoy of this decoy group has retired (its slot's "active" word is 0): nothing to evaluate and nobody reads the records --
  // the launches a host chunk has queued behind a fold's last report, and the folds a shared launch still names while they drain
  // (launch_engine.h), leave here instead of filling LDS and storing zeros.  The waves of a workgroup hold the same decoys: they
  // all leave or all stay (no barrier is left waiting).
  if (!__any((int)(live && w1.z != 0.0f))) return;
  for (int i = tix; i < ktot; i += NT) s_kn[i] = A.knots[i];
  // equal slices of the row's list: every workgroup of a row gets the same number of restraint visits.  The bounds come with the
  // work item (the host made the plan from the rows' lengths): reading the length here was one more dependent round trip in
  // front of the list itself.
  const int e_lo = (int)(item2.y & 0xffffu), e_hi = (int)(item2.y >> 16);
  for (int i = e_lo + tix; i < e_hi; i += NT) {
    s_ent[i - e_lo] = A.rows[(size_t)a * L + i];
    s_rx[i - e_lo] = A.rows_rx[(size_t)a * L + i];
  }
  if constexpr (NW == 1) wave_lds_sync(); else __syncthreads();
  const float* knd = s_kn;
  const float* kno = s_kn + kd;
  const float* knt = s_kn + kd + KO;
  const float* knp = s_kn + kd + 2 * KO;
  const float inv_o = 1.0f / (kno[1] - kno[0]), inv_p = 1.0f / (knp[1] - knp[0]);
  // distance knots: three unevenly spaced repulsive ones, then a uniform grid (0 / 2 / 3.5 / 4.25 + 0.5 k; AF2: 0 / 2.325 / 3.575 / 3.875 + 0.3125 k)
  const float kd1 = knd[1], kd2 = knd[2], kd3 = knd[3], inv_d = 1.0f / (knd[4] - knd[3]);

  const float w_ap = w0.x, w_dih = w0.y, w_ang = w0.z, w_vdw = w0.w, w_hb = w1.w;
  const int sep_lo = (int)w1.x, sep_hi = (int)w1.y;
  const bool active = live && w1.z != 0.0f;
  // which selection of the map's restraints this decoy's run sees: 1 + trx2_run.pair_filter (TRX2_FILTER_*)
  const int fsel = (int)w1.z - 1;

  // residue a
  f3 CAa, Na, CBa, Ca, Oa;
  xt_unpack(q0, q1, q2, q3, CAa, Na, CBa, Ca, Oa);
  const f3 Ha = mk3(q4.x, q4.y, q4.z);
  const bool donor_a = q4.w != 0.0f;

  f3 gN = mk3(0, 0, 0), gCA = gN, gC = gN, gO = gN, gCB = gN, gH = gN;
  float e_d = 0, e_o = 0, e_t = 0, e_p = 0, e_v = 0, e_h = 0;
  const unsigned aL = (unsigned)a * (unsigned)L;
  constexpr int VSTRIDE = NW * PW;
  STAMP(0)  // prologue: knots and list slice to LDS, barrier, weights, residue a

  // ---- (i) restraint terms: the slice of the row's list, PW entries per wave step (lane = decoy, sub-lane h = entry).  The
  // pair, hence the table row, is the same for all decoys of a sub-lane.  Skipped outright while no decoy of the wave has a
  // restraint weight or an open separation window (the declash runs, folding.py:119: no restraints loaded yet).
  constexpr bool use_c = SEGC;
  const bool want_rst = active && sep_hi > sep_lo && ((((FAM & FAM_DIST) != 0) && w_ap != 0.0f) || (((FAM & FAM_ANG) != 0) && (w_dih != 0.0f || w_ang != 0.0f)));
  if (__any((int)want_rst)) {
  // (Measured and not kept, profiles/README.md round 3: the visit as a lambda called from the loop -- the register allocation
  // of the all-channel instantiations got worse, 41 spilled registers at 16 decoys per wave -- and two entries per trip with
  // both entries' coordinates requested first: neutral with distances only, 10-25 % slower with all channels.)
#pragma unroll 1
  for (int e0 = e_lo + wave * PW; e0 < e_hi; e0 += VSTRIDE) {
    const int e = e0 + h;
    const unsigned ent = s_ent[min(e, e_hi - 1) - e_lo];
    const unsigned erx = s_rx[min(e, e_hi - 1) - e_lo];
    const int bc = (int)(ent & PAIR_ROW_B_BITS);
    // residue b's CA, N, CB: requested as soon as the entry is read, before its masks are examined
    const float4* xb = A.xyzT + (__umul24((unsigned)(grp * L + bc), 5u * BW) + (unsigned)d);
    const float4 r0 = xb[0], r1 = xb[BW], r2 = xb[2 * BW];
    // ... and with them this (entry, decoy)'s segment-cache block (PairArgs): its address does not wait for the geometry
    constexpr int NSEG = (FAM & FAM_ANG) ? 6 : 1;
    const size_t ci = (size_t)a * (size_t)A.rowcap + (size_t)min(e, e_hi - 1);   // (one decoy: the block belongs to the row entry)
    uint2 ctag = make_uint2(0xffffffffu, 0xffffffffu);
    float4 cseg[NSEG];
#pragma unroll
    for (int q = 0; q < NSEG; q++) cseg[q] = make_float4(0, 0, 0, 0);
    if constexpr (use_c) {
      ctag = A.segt[ci];
#pragma unroll
      for (int q = 0; q < NSEG; q++) cseg[q] = A.segc[ci * NSEG + q];
    }
    const int sep = abs(a - bc);
    unsigned m_ab = 0, m_ba = 0;
    if (want_rst && e < e_hi && sep >= sep_lo && sep < sep_hi) {
      const unsigned mm = (fsel == TRX2_FILTER_ODR && A.has_odr) ? (ent >> 24)
                          : fsel == TRX2_FILTER_RELAX1 ? (erx & 0xffu) : fsel == TRX2_FILTER_RELAX2 ? (erx >> 8) : ((ent >> 16) & 0xffu);
      m_ab = mm & 15u;
      m_ba = mm >> 4;
    }
    if (!(FAM & FAM_DIST)) { m_ab &= ~TRX2_M_DIST; m_ba &= ~TRX2_M_DIST; }
    if (!(FAM & FAM_ANG)) { m_ab &= TRX2_M_DIST; m_ba &= TRX2_M_DIST; }
    const unsigned msym = (a < bc) ? m_ab : m_ba;  // DIST / OMEGA bits live on the (min,max) row
    STAMP(1)  // entry + masks + loop control
    if (!__any((int)(m_ab | m_ba))) continue;

    const f3 CAb = mk3(r0.x, r0.y, r0.z), Nb = mk3(r0.w, r1.x, r1.y), CBb = mk3(r1.z, r1.w, r2.x);
    STAMP(2)  // coordinates of residue b (3 x 16 B per lane)
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
    const bool on_d = (FAM & FAM_DIST) && (msym & TRX2_M_DIST);
    f3 ud = u;
    float idd = iu, dd = du;
    if (A.dist_ca) { ud = CAa - CAb; const float d2 = dot(ud, ud); idd = frsq(d2); dd = d2 * idd; }
    SplSeg sd;
    float2 kd0 = make_float2(0, 0), kd1_ = kd0;
    bool miss_d = false, miss_a = false;
    if (on_d) {
      sd = spline_seek(knd, kd, dd < kd1 ? 0 : (dd < kd2 ? 1 : (dd < kd3 ? 2 : 3 + (int)((dd - kd3) * inv_d))), dd);
      seg_fetch(use_c, ctag.x & 0xffu, sd.idx, cseg[0], A.Td + __umul24(isym, (unsigned)kd) + sd.idx, kd0, kd1_, miss_d);
    }
    STAMP(3)  // dist: seek, fetch
    if constexpr ((FAM & FAM_ANG) != 0) {
    const unsigned m_om = msym & TRX2_M_OMEGA;
    const unsigned m_tp_ab = m_ab & (TRX2_M_THETA | TRX2_M_PHI), m_tp_ba = m_ba & (TRX2_M_THETA | TRX2_M_PHI);
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
      float2 o0, o1, t10, t11, t20, t21, p10, p11, p20, p21;
      if constexpr (!use_c) {   // (straight-line: the five gathers leave together)
        o0 = r_o[0]; o1 = r_o[1]; t10 = r_t1[0]; t11 = r_t1[1]; t20 = r_t2[0]; t21 = r_t2[1]; p10 = r_p1[0]; p11 = r_p1[1]; p20 = r_p2[0]; p21 = r_p2[1];
      } else {
        constexpr int Q = NSEG > 1 ? 1 : 0;   // (NSEG == 6 here: the angular block exists only with FAM_ANG)
        bool m1 = false, m2 = false, m3 = false, m4 = false, m5 = false;
        seg_fetch(true, (ctag.x >> 8) & 0xffu, g_o.idx, cseg[1 * Q], r_o, o0, o1, m1);
        seg_fetch(true, (ctag.x >> 16) & 0xffu, g_t1.idx, cseg[2 * Q], r_t1, t10, t11, m2);
        seg_fetch(true, ctag.x >> 24, g_t2.idx, cseg[3 * Q], r_t2, t20, t21, m3);
        seg_fetch(true, ctag.y & 0xffu, g_p1.idx, cseg[4 * Q], r_p1, p10, p11, m4);
        seg_fetch(true, (ctag.y >> 8) & 0xffu, g_p2.idx, cseg[5 * Q], r_p2, p20, p21, m5);
        if (m1 | m2 | m3 | m4 | m5) {   // refresh the block with the segments just gathered
          miss_a = true;
          float4* cb = A.segc + ci * NSEG;
          if (m1) cb[1 * Q] = make_float4(o0.x, o0.y, o1.x, o1.y);
          if (m2) cb[2 * Q] = make_float4(t10.x, t10.y, t11.x, t11.y);
          if (m3) cb[3 * Q] = make_float4(t20.x, t20.y, t21.x, t21.y);
          if (m4) cb[4 * Q] = make_float4(p10.x, p10.y, p11.x, p11.y);
          if (m5) cb[5 * Q] = make_float4(p20.x, p20.y, p21.x, p21.y);
          ctag.x = (ctag.x & 0xffu) | ((unsigned)g_o.idx << 8) | ((unsigned)g_t1.idx << 16) | ((unsigned)g_t2.idx << 24);
          ctag.y = (ctag.y & 0xffff0000u) | (unsigned)g_p1.idx | ((unsigned)g_p2.idx << 8);
        }
      }
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
    }
    if (on_d) {
      float ev, de;
      spline_value(sd, kd0, kd1_, dd, ev, de);
      if (first) e_d += ev;
      if (A.dist_ca) gCA = fma3(ud, w_ap * de * idd, gCA);
      else gCB = fma3(ud, w_ap * de * idd, gCB);
    }
    if constexpr (use_c) {
      if (miss_d) {   // (only lanes with the term on and a real entry reach a fetch, so only they write)
        A.segc[ci * NSEG] = make_float4(kd0.x, kd0.y, kd1_.x, kd1_.y);
        ctag.x = (ctag.x & 0xffffff00u) | (unsigned)sd.idx;
      }
      if (miss_d | miss_a) A.segt[ci] = ctag;
    }
    STAMP(8)  // values, gradients
  }
  }
  STAMP(9)  // restraint list done

  // ---- (ii) contacts: this workgroup's b-range of the row (the ranges of a row's workgroups tile 0..L-1).  Which residues
  // touch depends on the decoy, so in lockstep the 25 atom pairs ran whenever ANY of the 64 decoys was within the cutoff -- on
  // ~85 % of the visits of a distance-only fold although ~10 % of (pair, decoy) combinations are in contact
  // (profiles/README.md).  A scan over the partners' C-alpha records one contact bit per lane and visit, the loads of eight
  // visits in flight together; after a block of up to 32 visits every lane walks ITS OWN bits, gathering its own residue b.
  // The walk takes max-over-lanes(contacts) steps.  Order per lane stays fixed: deterministic.
  if constexpr ((FAM & FAM_VDW) != 0) {
  const bool want_vdw = active && (w_vdw != 0.0f || w_hb != 0.0f);
  if (__any((int)want_vdw)) {
  const int chunk = (L + nsl - 1) / nsl;
  const int b_lo = split * chunk, b_hi = min(L, b_lo + chunk);
  const float4* xg = A.xyzT + (__umul24((unsigned)(grp * L), 5u * BW) + (unsigned)d);
  for (int bb = b_lo + wave * PW; bb < b_hi; bb += 32 * VSTRIDE) {
    unsigned vmask = 0;
#pragma unroll 1
    for (int v0 = 0; v0 < 32; v0 += 8) {
      if (bb + v0 * VSTRIDE >= b_hi) break;
      float4 c[8];
#pragma unroll
      for (int q = 0; q < 8; q++) c[q] = xg[__umul24((unsigned)min(bb + (v0 + q) * VSTRIDE + h, L - 1), 5u * BW)];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int b = bb + (v0 + q) * VSTRIDE + h;
        const f3 dca = CAa - mk3(c[q].x, c[q].y, c[q].z);
        if (want_vdw && b < b_hi && abs(a - b) >= TRX2_VDW_MINSEP && dot(dca, dca) < (float)TRX2_VDW_CUT2) vmask |= 1u << (v0 + q);
      }
    }
    STAMP(10)  // contact scan
    // per-lane walk: every lane follows its own contact bits (max-over-lanes steps): repulsion over the 5 x 5 atom pairs and
    // the two hydrogen-bond candidates of the pair, gradient on residue a's atoms; symmetric energies counted from the lower row
    // (Round 4, measured and not kept: the NEXT contact's residue requested before the current one is evaluated -- 20 more live
    // registers: at three waves per SIMD every instantiation spilled 20-40 of them, at two the occupancy cost more than the overlap
    // gave: config 2 326 -> 270 / 318 decoys/s, pooled queue 1664 -> 1480 / 1437, config 3 637 -> 588 / 575; profiles/README.md.)
    while (vmask) {  // per-lane trip count; lanes without further contacts idle
      const int v = __ffs((int)vmask) - 1;
      vmask &= vmask - 1;
      const int b = bb + v * VSTRIDE + h;
      const float4* xb = xg + __umul24((unsigned)b, 5u * BW);
      const float4 r0 = xb[0], r1 = xb[BW], r2 = xb[2 * BW], r3 = xb[3 * BW], r4 = xb[4 * BW];
      f3 CAb, Nb, CBb, Cb, Ob;
      xt_unpack(r0, r1, r2, r3, CAb, Nb, CBb, Cb, Ob);
      const f3 Hb = mk3(r4.x, r4.y, r4.z);
      if (w_vdw != 0.0f) {
        const f3 pa[5] = {Na, CAa, Ca, Oa, CBa};
        const f3 pb[5] = {Nb, CAb, Cb, Ob, CBb};
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
    STAMP(11)  // contact walk
  }
  }
  }

  // ---- reduce over the PW residue sub-lanes (inside the wave) and over the waves (through LDS); write decoy-major records.
  // The sub-lanes of a decoy sit BW lanes apart: a butterfly over the lane distances BW, 2 BW, .. 32 leaves their sum in
  // every one of them.  (Summing all PAIR_WAVES * PW slots from LDS instead made 6 threads add 256 slots each when one
  // decoy is folded -- the case of every feedback iteration.)
  // LDS image [wave][decoy][21]: the h = 0 lanes write 20 values at stride 21 (no bank conflict); the readers are
  // (decoy, quad) pairs, 4 lanes per decoy, so every store instruction writes whole 64-B (gradient) / 32-B (energy) runs.
  {
    float vals[PR_REC] = {gN.x, gN.y, gN.z, gCA.x, gCA.y, gCA.z, gC.x, gC.y, gC.z, gO.x, gO.y, gO.z,
                           gCB.x, gCB.y, gCB.z, gH.x, gH.y, gH.z, e_d, e_o, e_t, e_p, e_v, e_h};
    // lane distances below 16: rotations inside the rows of 16 lanes (DPP row_ror, VALU speed; every lane of a decoy ends with
    // the row's sum, in its own order of additions -- only the h = 0 lane is used); 16 and 32: ds_bpermute
#pragma unroll
    for (int o = BW; o < 16; o <<= 1)
#pragma unroll
      for (int k = 0; k < PR_REC; k++)
        vals[k] += __int_as_float(o == 1 ? __builtin_amdgcn_update_dpp(0, __float_as_int(vals[k]), 0x121, 0xF, 0xF, false)
                                  : o == 2 ? __builtin_amdgcn_update_dpp(0, __float_as_int(vals[k]), 0x122, 0xF, 0xF, false)
                                  : o == 4 ? __builtin_amdgcn_update_dpp(0, __float_as_int(vals[k]), 0x124, 0xF, 0xF, false)
                                           : __builtin_amdgcn_update_dpp(0, __float_as_int(vals[k]), 0x128, 0xF, 0xF, false));
    if constexpr (NW != 1) {
#pragma unroll
    for (int o = (BW < 16 ? 16 : BW); o < 64; o <<= 1)
#pragma unroll
      for (int k = 0; k < PR_REC; k++) vals[k] += __shfl_xor(vals[k], o, 64);
    }
    if constexpr (NW == 1) {
      // (one wave, one decoy: BW == 1, so the loops above left every lane with the sum of its row of 16 lanes) the four rows by
      // v_readlane in a fixed order; lane 0 stores the record
      float tot[PR_REC];
#pragma unroll
      for (int k = 0; k < PR_REC; k++) tot[k] = (lane_value(vals[k], 0) + lane_value(vals[k], 16)) + (lane_value(vals[k], 32) + lane_value(vals[k], 48));
      STAMP(13)
      if (lane == 0 && live) {
        float4* o = reinterpret_cast<float4*>(A.FA + (((size_t)split * A.B + dec) * L + a) * PR_REC);
#pragma unroll
        for (int q = 0; q < 6; q++) o[q] = make_float4(tot[q * 4], tot[q * 4 + 1], tot[q * 4 + 2], tot[q * 4 + 3]);
      }
    } else {
    if (h == 0) {
      float* s = s_red + ((size_t)wave * BW + d) * RED_STRIDE;
#pragma unroll
      for (int k = 0; k < PR_REC; k++) s[k] = vals[k];
    }
    }
  }
  if constexpr (NW != 1) {
  STAMP(13)  // epilogue: sums over the sub-lanes, LDS image
  __syncthreads();
  STAMP(14)  // epilogue: barrier (the other waves of the workgroup)
  for (int t = threadIdx.x; t < 6 * BW; t += NT) {  // 6 quads per decoy = one 24-float record
    const int dd = t / 6, q = t % 6;
    const int dc = grp * BW + dd;
    if (dc >= A.B) continue;
    float acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int sl = 0; sl < NW; sl++) acc[i] += s_red[((size_t)sl * BW + dd) * RED_STRIDE + q * 4 + i];
    reinterpret_cast<float4*>(A.FA + (((size_t)split * A.B + dc) * L + a) * PR_REC)[q] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
  }
  STAMP(12)  // epilogue: LDS image, barrier, column sums, stores
  STAMP_FLUSH
}
template <int BW, int FAM>
__global__ __launch_bounds__(PAIR_THREADS, (pair_min_waves<BW, FAM>())) void k_pair(PairArgs A) {
  pair_body<BW, FAM>(A, blockIdx.x, (int)blockIdx.z);
}
// Single-decoy folds with the segment cache: 24 more live registers in the all-channel visit -- two waves per SIMD there (at three
// the cache variants spill 70-90 registers: measured 1.5 x slower), three with distances only.
template <int FAM>
constexpr int pair_c_min_waves() { return (FAM & FAM_ANG) ? 2 : PAIR_MIN_WAVES_DIST; }
template <int FAM>
__global__ __launch_bounds__(PAIR_THREADS, (pair_c_min_waves<FAM>())) void k_pair_c(PairArgs A) {   // four waves per row
  pair_body<1, FAM, PAIR_WAVES, true>(A, blockIdx.x, (int)blockIdx.z);
}
// ... one wave per work item (row), 64-thread workgroups
template <int FAM, bool SEGC>
__global__ __launch_bounds__(64, (SEGC ? pair_c_min_waves<FAM>() : pair_min_waves<1, FAM>())) void k_pair1(PairArgs A) {
  pair_body<1, FAM, 1, SEGC>(A, blockIdx.x, (int)blockIdx.z);
}
// Shared launch (trx2fold.hip: LaunchEngine): ONE launch evaluates the pair terms of several independent folds, each on its own
// map -- blockIdx.z picks the fold, whose argument block (tables, row lists, row plan, coordinates, records, chain length) is read
// from device memory instead of the kernel arguments.  A fold here is one decoy (the iteration phase of run_inference.py folds
// one decoy per map and iteration, run_inference.py:97-139: a single-decoy launch leaves the chip idle, and four streams are all
// the hardware queues there are).  The body is k_pair1's: a fold's arithmetic does not depend on what shares its launch.
// an argument block from device memory through the CONSTANT address space: scalar loads into SGPRs, exactly what the kernel
// arguments of the single-fold kernels are (the blocks are uploaded before the launch and never written during it)
template <class T>
__device__ __forceinline__ T load_args(const T* p) {
  static_assert(sizeof(T) % 4 == 0, "argument blocks are copied word by word");
  typedef __attribute__((address_space(4))) const unsigned* cword;
  const cword src = (cword)(unsigned long long)p;
  union U { T t; unsigned w[sizeof(T) / 4]; __device__ U() {} } u;
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; i++) u.w[i] = src[i];
  return u.t;
}
// the four-wave workgroups in a shared launch (contexts whose single-decoy folds keep four waves per row)
template <int FAM, bool SEGC>
__global__ __launch_bounds__(PAIR_THREADS, (SEGC ? pair_c_min_waves<FAM>() : pair_min_waves<1, FAM>())) void k_pair_multi(const PairArgs* AA) {
  const PairArgs A = load_args(AA + blockIdx.z);
  if ((int)blockIdx.x >= A.n_items) return;
  pair_body<1, FAM, PAIR_WAVES, SEGC>(A, blockIdx.x, 0);
}
// Placement.  Both visits of a residue pair -- from row a and from row b -- read the same six spline segments, and a fold's rows
// read the same coordinate records; the table lines a fold touches in one evaluation are ~2 MB, half of an XCD's L2.  Workgroups are
// dealt round-robin over the eight XCDs (observed, never relied on for correctness: MI355X_MICROARCH.md, workgroup dispatch), whose
// L2s do not share lines.  With eight or more folds in a launch (xcd_groups != 0) the 1-D grid is laid out so that ALL rows of a fold
// carry the same block id modulo 8 -- one XCD, one L2: the second visit of a pair finds its segments there.  Measured at 14 folds
// per launch (profiles/README.md, round 4): L2 hit rate 29 % and 51 MB fetched per launch with the rows spread over the XCDs.
// With fewer folds a fold's rows stay spread: one XCD is an eighth of the chip.  The mapping changes no result.
template <int FAM, bool SEGC>
__global__ __launch_bounds__(64, (SEGC ? pair_c_min_waves<FAM>() : pair_min_waves<1, FAM>())) void k_pair1_multi(const PairArgs* AA, int n_folds, int max_items, int xcd_groups) {
  unsigned fold, item;
  if (xcd_groups) {
    const unsigned id = blockIdx.x, x = id & 7u, s = id >> 3;
    fold = x + 8u * (s / (unsigned)max_items);
    item = s % (unsigned)max_items;
  } else {
    fold = blockIdx.x / (unsigned)max_items;
    item = blockIdx.x % (unsigned)max_items;
  }
  if ((int)fold >= n_folds) return;
  const PairArgs A = load_args(AA + fold);
  if ((int)item >= A.n_items) return;
  pair_body<1, FAM, 1, SEGC>(A, item, 0);
}
