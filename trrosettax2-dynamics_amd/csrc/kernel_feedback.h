// kernel_feedback.h -- K7: the feedback step between folds on the device -- included by trx2fold.hip.
// Not a stand-alone header: it relies on np_sum_f32_dev (kernel_tables.h) and the helpers above its #include.
#pragma once
// =================================================================================================
// K7: decoy -> realised 6-D geometry -> bins (utils_trX2dy/utils.py:125-235) and the re-weighting of a distogram
// channel with those bins (utils.py:379-403).  The host-side mirror (feedback.py) is bit-identical to the reference; these
// kernels repeat ITS arithmetic in numpy's operation order with no fused multiply-add: float64 geometry (on float32-valued
// coordinates, as the reference's reader hands them over), float32 for the distogram rows with numpy's pairwise sums,
// float64 inside the Gaussian filter as scipy does.  What cannot be promised bit for bit is atan2 (glibc vs the device
// library, last ulp): a value within one ulp of a bin edge may fall on the other side.  The parity test counts such pairs
// (none seen) and compares everything else exactly.
// =================================================================================================
struct FbBinsArgs {
  int L;
  const float* xyz;            // [L][5][3] N CA C O CB, NaN = absent
  const unsigned char* gly;    // [L] 1 = glycine (virtual C-beta)
  const double *d_edges, *a_edges, *p_edges;  // np.arange(2, 20.5, .5), np.arange(-pi, pi, pi/12), np.arange(0, pi, pi/12)
  int nd, na, np_;
  double dmax2;
  signed char *jd, *jo, *jt, *jp;  // [L][L]
};

// Geometry in float64 on float32-valued coordinates: the reference's reader fills float64 arrays with Biopython's float32
// coordinates (utils.py:270,283), so get_neighbors computes in float64.
struct v3d { double x, y, z; };
#pragma clang fp contract(off)
__device__ __forceinline__ v3d fb_sub(v3d a, v3d b) { return v3d{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ double fb_sum3(double a, double b, double c) { return (a + b) + c; }       // np.sum over an axis of 3
__device__ __forceinline__ double fb_dot(v3d a, v3d b) { return fb_sum3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3d fb_cross(v3d a, v3d b) {  // np.cross: multiply, multiply, subtract
  return v3d{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ double fb_norm(v3d a) { return __dsqrt_rn(fb_dot(a, a)); }                   // np.linalg.norm
// get_dihedrals (utils.py:97-110)
__device__ __forceinline__ double fb_dihedral(v3d a, v3d b, v3d c, v3d d) {
  v3d t = fb_sub(b, a);
  v3d b0 = v3d{-1.0 * t.x, -1.0 * t.y, -1.0 * t.z}, b1 = fb_sub(c, b), b2 = fb_sub(d, c);
  const double n = fb_norm(b1);
  b1 = v3d{b1.x / n, b1.y / n, b1.z / n};
  const double s0 = fb_dot(b0, b1), s2 = fb_dot(b2, b1);
  const v3d v = v3d{b0.x - s0 * b1.x, b0.y - s0 * b1.y, b0.z - s0 * b1.z};
  const v3d w = v3d{b2.x - s2 * b1.x, b2.y - s2 * b1.y, b2.z - s2 * b1.z};
  const double x = fb_dot(v, w), y = fb_dot(fb_cross(b1, v), w);
  return atan2(y, x);
}
__device__ __forceinline__ v3d fb_atom(const FbBinsArgs& A, int i, int k) {
  const float* p = A.xyz + (size_t)i * 15 + k * 3;
  return v3d{(double)p[0], (double)p[1], (double)p[2]};
}
__device__ __forceinline__ v3d fb_cbeta(const FbBinsArgs& A, int i) {
  const v3d N = fb_atom(A, i, 0), Ca = fb_atom(A, i, 1), C = fb_atom(A, i, 2), real = fb_atom(A, i, 4);
  const v3d b = fb_sub(Ca, N), c = fb_sub(C, Ca), a = fb_cross(b, c);
  const double k0 = -0.58273431, k1 = 0.56802827, k2 = -0.54067466;  // utils.py:135
  const v3d virt{((k0 * a.x + k1 * b.x) + k2 * c.x) + Ca.x, ((k0 * a.y + k1 * b.y) + k2 * c.y) + Ca.y, ((k0 * a.z + k1 * b.z) + k2 * c.z) + Ca.z};
  const bool use = !A.gly[i] && isfinite(real.x) && isfinite(real.y) && isfinite(real.z);
  return use ? real : virt;
}
// Reliability score of n decoys (calculate_reliability_score, utils_trX2dy/utils.py:352-372 over Biopython's PPBuilder phi/psi,
// :337-349): residues bonded on both sides (C-N peptide distance < 1.8 A) whose phi lies in [-180, 0] -- the reference compares
// RADIANS with those degrees, so the test is phi <= 0 (quirk R10) and the psi test is always true.  One thread per (decoy,
// residue); counts[2 d] = residues with both angles, counts[2 d + 1] = those with phi <= 0.  Same float64 arithmetic as above.
__global__ void k_reliability(int n, int L, const float* xyz /* [n][L][5][3] as read from the PDB files */, int* counts) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * L) return;
  const int d = t / L, i = t % L;
  if (i < 1 || i > L - 2) return;
  const float* p = xyz + (size_t)d * L * 15;
  auto at = [&](int r, int k) { const float* q = p + (size_t)r * 15 + k * 3; return v3d{(double)q[0], (double)q[1], (double)q[2]}; };
  const v3d Cp = at(i - 1, 2), N = at(i, 0), CA = at(i, 1), C = at(i, 2), Nn = at(i + 1, 0);
  if (!(fb_norm(fb_sub(Cp, N)) < 1.8) || !(fb_norm(fb_sub(C, Nn)) < 1.8)) return;
  const double phi = fb_dihedral(Cp, N, CA, C);
  atomicAdd(counts + 2 * d, 1);
  if (-180.0 <= phi && phi <= 0.0) atomicAdd(counts + 2 * d + 1, 1);
}

__device__ __forceinline__ int fb_count(const double* edges, int n, double x) {  // (edges < x).sum()
  int c = 0;
  for (int k = 0; k < n; k++) c += edges[k] < x;
  return c;
}
__global__ void k_fb_bins(FbBinsArgs A) {
  const size_t ij = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ij >= (size_t)A.L * A.L) return;
  const int i = (int)(ij / A.L), j = (int)(ij % A.L);
  int jd = 0, jo = 0, jt = 0, jp = 0;
  if (i != j) {
    const v3d cbi = fb_cbeta(A, i), cbj = fb_cbeta(A, j);
    const double dx = cbi.x - cbj.x, dy = cbi.y - cbj.y, dz = cbi.z - cbj.z;
    const double d2 = (dx * dx + dy * dy) + dz * dz;
    if (d2 <= A.dmax2) {
      const v3d Ni = fb_atom(A, i, 0), Cai = fb_atom(A, i, 1), Caj = fb_atom(A, j, 1);
      const double dist = fb_norm(fb_sub(cbj, cbi));
      jd = fb_count(A.d_edges, A.nd, dist);
      if (jd >= 37) jd = 0;
      if (jd != 0) {
        const double om = fb_dihedral(Cai, cbi, cbj, Caj), th = fb_dihedral(Ni, Cai, cbi, cbj);
        jo = fb_count(A.a_edges, A.na, om);
        jt = fb_count(A.a_edges, A.na, th);
        jp = fb_count(A.p_edges, A.np_, th);  // the reference bins THETA on phi's edges (utils.py:226)
      }
    }
  }
  A.jd[ij] = (signed char)jd; A.jo[ij] = (signed char)jo; A.jt[ij] = (signed char)jt; A.jp[ij] = (signed char)jp;
}

struct FbProcArgs {
  int L, K, norm, smooth;
  const float* in;           // [L][L][K]
  const signed char* bins;   // [L][L]
  double w[9];               // Gaussian weights, radius 4 (scipy _gaussian_kernel1d)
  float* out;                // [L][L][K]
  unsigned* max_diff_bits;   // optional: max |out - in| over everything, as the bits of a non-negative float (atomicMax)
};
// process_distribution_with_pred_distribution (utils.py:379-403): one thread per pair
__global__ void k_fb_process(FbProcArgs A) {
  const size_t ij = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ij >= (size_t)A.L * A.L) return;
  const int K = A.K;
  const float* src = A.in + ij * K;
  float* dst = A.out + ij * K;
  float row[40];
  float mx = src[0];
  for (int k = 0; k < K; k++) { row[k] = src[k]; mx = fmaxf(mx, src[k]); }
  if (!(mx < 0.5f)) {  // pairs the network is sure about are left alone
    for (int k = 0; k < K; k++) dst[k] = row[k];
    return;
  }
  if (A.max_diff_bits && !A.norm) {  // the convergence measure of the iteration: max |tmp_new - tmp_old| (run_inference.py:133)
    const int i0 = A.bins[ij];
    if (i0 < K - 1) {
      const float v = row[i0], nv = v < 0.05f ? v : v * 0.5f;
      atomicMax(A.max_diff_bits, __float_as_uint(fabsf(nv - v)));
    }
  }
  const int idx = A.bins[ij];
  if (idx < K - 1) {  // realised bin = last bin: the reference's slice is empty, no decay
    const float v = row[idx];
    row[idx] = v < 0.05f ? v : v * 0.5f;
  }
  if (!A.norm) {  // the cumulative `tmp` array: decayed, not normalised
    for (int k = 0; k < K; k++) dst[k] = row[k];
    return;
  }
  const float s = np_sum_f32_dev(row, K);
  for (int k = 0; k < K; k++) row[k] = row[k] / s;
  if (!A.smooth) {
    for (int k = 0; k < K; k++) dst[k] = row[k];
    return;
  }
  // scipy.ndimage.gaussian_filter1d(mode="reflect"): double line with 4 reflected samples on each side, symmetric
  // correlation accumulated from the outermost pair inwards (NI_Correlate1D), result cast to float32
  double line[48];
  for (int k = 0; k < K; k++) line[4 + k] = (double)row[k];
  for (int m = 1; m <= 4; m++) { line[4 - m] = (double)row[m - 1]; line[4 + K - 1 + m] = (double)row[K - m]; }
  for (int l = 0; l < K; l++) {
    const double* c = line + 4 + l;
    double t = c[0] * A.w[4];
    for (int jj = -4; jj < 0; jj++) t += (c[jj] + c[-jj]) * A.w[4 + jj];
    dst[l] = (float)t;
  }
}
#pragma clang fp contract(fast)

// =================================================================================================
// GloCon matrix for clustering (utils_trX2dy/utils.py:543-569): score(p, q) = sum over the upper triangle of
// |dist6d_p - dist6d_q| with differences <= 3 A dropped, divided by L (L - 1) / 2.
// =================================================================================================
#pragma clang fp contract(off)
struct GloArgs {
  int n, L;
  const float* xyz;          // [n][L][5][3]
  const unsigned char* gly;  // [n][L]
  double dmax2;
  double* d6;                // [n][L][L] dist6d of every decoy (0 outside dmax), float64 like the reference's arrays
  double* out;               // [n][n]
};
__global__ void k_glocon_dist(GloArgs A) {  // dist6d of every decoy: get_neighbors' first output
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t LL = (size_t)A.L * A.L;
  if (t >= (size_t)A.n * LL) return;
  const int p = (int)(t / LL), i = (int)((t % LL) / A.L), j = (int)(t % A.L);
  FbBinsArgs B;
  B.L = A.L; B.xyz = A.xyz + (size_t)p * A.L * 15; B.gly = A.gly + (size_t)p * A.L;
  double v = 0.0;
  if (i != j) {
    const v3d cbi = fb_cbeta(B, i), cbj = fb_cbeta(B, j);
    const double dx = cbi.x - cbj.x, dy = cbi.y - cbj.y, dz = cbi.z - cbj.z;
    if ((dx * dx + dy * dy) + dz * dz <= A.dmax2) v = fb_norm(fb_sub(cbj, cbi));
  }
  A.d6[t] = v;
}
// element k of np.triu(np.abs(d1 - d2) with values <= 3 zeroed), flattened
__device__ __forceinline__ double glo_elem(const double* a, const double* b, int L, int k) {
  const int i = k / L, j = k % L;
  if (j < i) return 0.0;
  const double d = fabs(a[k] - b[k]);
  return d <= 3.0 ? 0.0 : d;
}
// numpy's pairwise_sum over the flattened array: blocks of <= 128 elements with eight running sums; above that the range
// is halved (first half rounded down to a multiple of 8) and the two sums added.  Written with an explicit stack: device
// recursion makes the runtime reserve a dynamic stack for every wave slot (1.2 GB here).
__device__ __forceinline__ double glo_leaf(const double* a, const double* b, int L, int k0, int n) {
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; i++) r += glo_elem(a, b, L, k0 + i);
    return r;
  }
  double r[8];
  for (int q = 0; q < 8; q++) r[q] = glo_elem(a, b, L, k0 + q);
  int i;
  for (i = 8; i < n - (n % 8); i += 8)
    for (int q = 0; q < 8; q++) r[q] += glo_elem(a, b, L, k0 + i + q);
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; i++) res += glo_elem(a, b, L, k0 + i);
  return res;
}
__device__ double glo_pairwise(const double* a, const double* b, int L, int k0_, int n_) {
  int fk0[24], fn[24], fstate[24];  // depth <= log2(4096^2 / 128) + 1 = 18
  double fleft[24];
  int sp = 0;
  fk0[0] = k0_; fn[0] = n_; fstate[0] = 0;
  double ret = 0.0;
  while (sp >= 0) {
    const int k0 = fk0[sp], n = fn[sp];
    if (n <= 128) { ret = glo_leaf(a, b, L, k0, n); sp--; continue; }
    int n2 = n / 2;
    n2 -= n2 % 8;
    if (fstate[sp] == 0) {        // descend into the first half
      fstate[sp] = 1;
      sp++; fk0[sp] = k0; fn[sp] = n2; fstate[sp] = 0;
    } else if (fstate[sp] == 1) { // first half done: keep it, descend into the second
      fleft[sp] = ret;
      fstate[sp] = 2;
      sp++; fk0[sp] = k0 + n2; fn[sp] = n - n2; fstate[sp] = 0;
    } else {                      // both done
      ret = fleft[sp] + ret;
      sp--;
    }
  }
  return ret;
}
__global__ void k_glocon_pairs(GloArgs A) {  // one thread per ordered pair p > q; the matrix is symmetric, diagonal 0
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)A.n * A.n) return;
  const int p = (int)(t / A.n), q = (int)(t % A.n);
  if (p <= q) return;
  const size_t LL = (size_t)A.L * A.L;
  const double s = glo_pairwise(A.d6 + (size_t)p * LL, A.d6 + (size_t)q * LL, A.L, 0, (int)LL);
  const double score = s / ((double)A.L * (double)(A.L - 1) / 2.0);
  A.out[(size_t)p * A.n + q] = score;
  A.out[(size_t)q * A.n + p] = score;
}
#pragma clang fp contract(fast)
