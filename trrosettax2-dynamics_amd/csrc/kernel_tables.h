// kernel_tables.h -- K2: restraint-table build (gen_rst + add_rst selection) -- included by trx2fold.hip.
// Not a stand-alone header: it relies on the macros, constant tables and helpers defined above its #include.
#pragma once
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
  // gen_idp_rst (utils_ros.py:196-373): on pairs flagged in idr the tables are normalised by the pair's most probable bin
  const unsigned char* idr;  // [L][L] or NULL
  int kind;                  // 0 gen_rst, 1 gen_idp_rst
  const double* idr_bk;      // [32][32] (bins_k / bins_m)^ALPHA, host pow (see bkgr)
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
  double y2[TRX2_KD_AF2], u[TRX2_KD_AF2];
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
  const bool flagged = A.kind == 1 && A.idr && A.idr[ab];
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
      if (flagged) {  // utils_ros.py:251-252: background relative to the bin of the maximum, normalised by the maximum
        int km = 0;
        _Pragma("unroll 1") for (int k = 1; k < 32; k++) if (row[5 + k] > row[5 + km]) km = k;  // np.argmax: first maximum
        _Pragma("unroll 1") for (int k = 0; k < 32; k++) {
          float num = row[5 + k] + meff32;
          double den = (double)row[5 + km] * A.idr_bk[km * 32 + k] + 1e-6;
          y[3 + k] = round_dp(-log((double)num / den) + A.ebase, 1e3);
        }
      }
      double rep0 = attr0 > 0.0 ? attr0 : 0.0;  // the repulsive knots keep the last-bin normalisation (:255)
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
      if (flagged) {  // utils_ros.py:285,312
        float mx = row[0];
        _Pragma("unroll 1") for (int k = 1; k < TRX2_NO_BINS; k++) if (row[k] > mx) mx = row[k];
        den = mx + meff32;
      }
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
        if (flagged) {  // utils_ros.py:348
          float mx = row[0];
          _Pragma("unroll 1") for (int k = 1; k < TRX2_NP_BINS; k++) if (row[k] > mx) mx = row[k];
          den = mx + meff32;
        }
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

// gen_rst_af2 (utils_ros.py:148-194): AlphaFold-style distogram dist[L][L][64]; 60 knots (A.knots: 3 repulsive + the bin edges
// 5..61), generation threshold 0.0025, the LAST bin's background on every bin (:172), normalised by bin 62; restraint on C-alpha
struct BuildAf2Args {
  int L;
  const float* dist;
  double ebase, erep[3], meff, pcut, bk_last;
  const double* knots;  // [60]
  float2* Td;           // [L][L][60]
  float* pd;
  unsigned char *gen, *sel;
};
__global__ void k_build_tables_af2(BuildAf2Args A) {
  const int L = A.L;
  size_t ab = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ab >= (size_t)L * L) return;
  const int a = (int)(ab / L), b = (int)(ab % L);
  const float* row = A.dist + ab * 64;
  constexpr int NA = TRX2_KD_AF2 - 3;
  const float p = np_sum_f32_dev(row + 6, NA);
  A.pd[ab] = p;
  unsigned char gen = 0, sel = 0;
  if ((double)p > 0.0025 && b > a) {
    double y[TRX2_KD_AF2], attr0 = 0;
    const float meff32 = (float)A.meff;
    _Pragma("unroll 1") for (int k = 0; k < NA; k++) {
      float num = row[6 + k] + meff32;
      double den = (double)row[62] * A.bk_last + 1e-6;
      double at = -log((double)num / den) + A.ebase;
      if (k == 0) attr0 = at;
      y[3 + k] = round_dp(at, 1e3);
    }
    const double rep0 = attr0 > 0.0 ? attr0 : 0.0;
    _Pragma("unroll 1") for (int k = 0; k < 3; k++) y[k] = round_dp(rep0 + A.erep[k], 1e3);
    spline_store(TRX2_KD_AF2, A.knots, y, A.Td + ab * TRX2_KD_AF2);
    gen = TRX2_M_DIST;
    if ((double)p >= A.pcut) sel = TRX2_M_DIST;
  }
  A.gen[ab] = gen;
  A.sel[ab] = sel;
}

// replace the values of table rows (gen_gpcr_rst's edits of the idr pairs): one thread per row, second derivatives recomputed
__global__ void k_override_rows(int n, int K, int L, const int* a, const int* b, const double* y, const double* knots, float2* T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double yy[TRX2_KD_AF2];
  _Pragma("unroll 1") for (int k = 0; k < K; k++) yy[k] = y[(size_t)i * K + k];
  spline_store(K, knots, yy, T + ((size_t)a[i] * L + b[i]) * K);
}

// mask2[a][b] = sel[a][b] | sel[b][a] << 4 : both directions of an ordered pair in ONE row-contiguous byte.  k_pair used
// to fetch sel[a][b] and sel[b][a] (a column access: a fresh cache line per visit) before it could even issue its
// coordinate loads -- ~950 of ~3250 cycles per visit (s_memtime stamps, profiles/README.md).
// mask_odr: the same with the restraints of the pairs flagged in idr removed (add_idr_rst with the complement mask, mode 3's
// first stage, folding.py:173-179), or NULL when the map has no idr mask.
__global__ void k_pack_masks(int L, const unsigned char* sel, const unsigned char* idr, unsigned char* mask2, unsigned char* mask_odr) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)L * L) return;
  const int a = (int)(i / L), b = (int)(i % L);
  const size_t j = (size_t)b * L + a;
  mask2[i] = (unsigned char)((sel[i] & 15) | ((sel[j] & 15) << 4));
  if (mask_odr) mask_odr[i] = (unsigned char)((idr[i] ? 0 : (sel[i] & 15)) | ((idr[j] ? 0 : (sel[j] & 15)) << 4));
}

// The relax stage's two re-selections, packed like mask2 (folding.py:230-237: add_rst(.., nogly=True) at PCUT 0.15, then 0.30;
// utils_ros.py:713-717): of the restraints gen_rst GENERATED, those with p >= pcut (dist), pcut + 0.5 (omega, theta),
// pcut + 0.6 (phi) whose two residues are not glycine.  po / pt / pp are NULL for maps without the angle channels.
struct RelaxSelArgs {
  int L;
  const unsigned char* gen;
  const float *pd, *po, *pt, *pp;
  const unsigned char* gly;   // [L] residue is a glycine
  double pc1, pc2;
  unsigned char *mask_r1, *mask_r2;
};
__device__ __forceinline__ unsigned relax_bits(const RelaxSelArgs& A, size_t ab, double pc) {
  const unsigned g = A.gen[ab];
  unsigned m = 0;
  if ((g & TRX2_M_DIST) && (double)A.pd[ab] >= pc) m |= TRX2_M_DIST;
  if (A.po && (g & TRX2_M_OMEGA) && (double)A.po[ab] >= pc + 0.5) m |= TRX2_M_OMEGA;
  if (A.pt && (g & TRX2_M_THETA) && (double)A.pt[ab] >= pc + 0.5) m |= TRX2_M_THETA;
  if (A.pp && (g & TRX2_M_PHI) && (double)A.pp[ab] >= pc + 0.6) m |= TRX2_M_PHI;
  return m;
}
__global__ void k_relax_masks(RelaxSelArgs A) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)A.L * A.L) return;
  const int a = (int)(i / A.L), b = (int)(i % A.L);
  const size_t j = (size_t)b * A.L + a;
  const bool ok = a != b && !A.gly[a] && !A.gly[b];
  A.mask_r1[i] = ok ? (unsigned char)(relax_bits(A, i, A.pc1) | (relax_bits(A, j, A.pc1) << 4)) : 0;
  A.mask_r2[i] = ok ? (unsigned char)(relax_bits(A, i, A.pc2) | (relax_bits(A, j, A.pc2) << 4)) : 0;
}
