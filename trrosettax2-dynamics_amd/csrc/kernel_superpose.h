// kernel_superpose.h -- batched optimal superposition of decoys: C-alpha RMSD and TM-score matrices -- included
// by trx2fold.hip.  Replaces the per-pair `TMscore` subprocess of /root/reference/utils_trX2dy/utils.py:514-541 (clustering) and
// evaluate_utils.py:33-100 (evaluation); the host mirror is trrosettax2-dynamics_amd/evaluate.py (rmsd_common, tm_score).
#pragma once
// Superposition without an SVD: the rotation that best maps x onto y is the eigenvector of the largest eigenvalue of Horn's
// symmetric 4 x 4 matrix N(S), S = sum (x - cx)(y - cy)^T; it is always a proper rotation, and
// RMSD^2 = (sum |x - cx|^2 + |y - cy|^2 - 2 lambda_max) / n.  The 4 x 4 problem is solved by cyclic Jacobi rotations in float64
// (every lane does it redundantly on wave-uniform data; it is ~1 % of an iteration).  Agreement with numpy's SVD route: 1e-14.
__device__ __forceinline__ void jacobi4_max(double (&A)[4][4], double& lam, double (&q)[4]) {
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  double tr = 0;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) tr += A[i][j] * A[i][j];
#pragma unroll 1
  for (int sweep = 0; sweep < 30; sweep++) {
    double off = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = i + 1; j < 4; j++) off += A[i][j] * A[i][j];
    if (off <= 1e-30 * tr) break;
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int r = p + 1; r < 4; r++) {
        const double apq = A[p][r];
        if (fabs(apq) < 1e-300) continue;
        const double th = (A[r][r] - A[p][p]) / (2.0 * apq);
        const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
        for (int k = 0; k < 4; k++) {  // A <- A J
          const double akp = A[k][p], akq = A[k][r];
          A[k][p] = c * akp - s * akq; A[k][r] = s * akp + c * akq;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {  // A <- J^T A
          const double apk = A[p][k], aqk = A[r][k];
          A[p][k] = c * apk - s * aqk; A[r][k] = s * apk + c * aqk;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const double vkp = V[k][p], vkq = V[k][r];
          V[k][p] = c * vkp - s * vkq; V[k][r] = s * vkp + c * vkq;
        }
      }
  }
  int best = 0;
#pragma unroll
  for (int i = 1; i < 4; i++) if (A[i][i] > A[best][best]) best = i;
  lam = A[best][best];
#pragma unroll
  for (int k = 0; k < 4; k++) q[k] = V[k][best];
}

// sums over a (sub)set of residue pairs -> rotation M (x' = M (x - cx) + cy), centres, largest eigenvalue
struct Sup { double M[3][3], cx[3], cy[3], lam, e0, cnt; };
__device__ __forceinline__ Sup sup_from_sums(const double (&v)[17]) {
  // v: cnt, sx[3], sy[3], sxy[9] (row-major x_a y_b), e0 = sum |x|^2 + |y|^2
  Sup R;
  R.cnt = v[0];
  const double ic = 1.0 / v[0];
  double S[3][3];
#pragma unroll
  for (int a = 0; a < 3; a++) { R.cx[a] = v[1 + a] * ic; R.cy[a] = v[4 + a] * ic; }
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) S[a][b] = v[7 + a * 3 + b] - v[1 + a] * v[4 + b] * ic;
  R.e0 = v[16] - (v[1] * v[1] + v[2] * v[2] + v[3] * v[3] + v[4] * v[4] + v[5] * v[5] + v[6] * v[6]) * ic;
  double N[4][4] = {{S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
                    {S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
                    {S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
                    {S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]}};
  double q[4];
  jacobi4_max(N, R.lam, q);
  const double q0 = q[0], qx = q[1], qy = q[2], qz = q[3];
  R.M[0][0] = q0 * q0 + qx * qx - qy * qy - qz * qz; R.M[0][1] = 2 * (qx * qy - q0 * qz); R.M[0][2] = 2 * (qx * qz + q0 * qy);
  R.M[1][0] = 2 * (qy * qx + q0 * qz); R.M[1][1] = q0 * q0 - qx * qx + qy * qy - qz * qz; R.M[1][2] = 2 * (qy * qz - q0 * qx);
  R.M[2][0] = 2 * (qz * qx - q0 * qy); R.M[2][1] = 2 * (qz * qy + q0 * qx); R.M[2][2] = q0 * q0 - qx * qx - qy * qy + qz * qz;
  return R;
}

// pair index -> (i, j).  Rectangular: row-major over n x m.  Symmetric: the k-th pair of the upper triangle i <= j, row by
// row (row i holds n - i pairs and starts at i n - i (i - 1) / 2): only those pairs are launched, the host mirrors them.
__device__ __forceinline__ void sup_pair_ij(long k, int n, int m, int symmetric, int& i, int& j) {
  if (!symmetric) { i = (int)(k / m); j = (int)(k % m); return; }
  const double b = 2.0 * n + 1.0;
  long r = (long)((b - sqrt(b * b - 8.0 * (double)k)) * 0.5);
  if (r < 0) r = 0;
  if (r > n - 1) r = n - 1;
  auto start = [&](long q) { return q * n - q * (q - 1) / 2; };
  while (r > 0 && start(r) > k) r--;
  while (r < n - 1 && start(r + 1) <= k) r++;
  i = (int)r; j = (int)(r + (k - start(r)));
}
struct SupArgs {
  int n, m, L, nseed, symmetric;
  long npair;          // pairs launched: n * m, or n (n + 1) / 2 when symmetric
  const float* xa;     // [n][L][3] C-alpha coordinates (as read from the PDB files)
  const float* xb;     // [m][L][3]
  const int2* seeds;   // [nseed] (start, length) of the TM-score program's seed fragments
  double lnorm, d0, d0_search;
  double* rmsd;                 // [n][m] or NULL
  unsigned long long* tm_bits;  // [n][m] best score as the bit pattern of a non-negative double (atomicMax), or NULL
};

// ---- RMSD: one wave per pair
__global__ __launch_bounds__(256) void k_sup_rmsd(SupArgs A) {
  const int lane = threadIdx.x & 63;
  const long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= A.npair) return;
  int i, j;
  sup_pair_ij(pair, A.n, A.m, A.symmetric, i, j);  // symmetric: i <= j only, mirrored by the host
  const float* x = A.xa + (size_t)i * A.L * 3;
  const float* y = A.xb + (size_t)j * A.L * 3;
  double v[17];
#pragma unroll
  for (int k = 0; k < 17; k++) v[k] = 0;
  for (int r = lane; r < A.L; r += 64) {
    const double px[3] = {x[r * 3], x[r * 3 + 1], x[r * 3 + 2]}, py[3] = {y[r * 3], y[r * 3 + 1], y[r * 3 + 2]};
    v[0] += 1.0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      v[1 + a] += px[a]; v[4 + a] += py[a]; v[16] += px[a] * px[a] + py[a] * py[a];
#pragma unroll
      for (int b = 0; b < 3; b++) v[7 + a * 3 + b] += px[a] * py[b];
    }
  }
#pragma unroll
  for (int k = 0; k < 17; k++) v[k] = wave_sum(v[k]);
  const Sup R = sup_from_sums(v);
  if (lane == 0) {
    const double ms = (R.e0 - 2.0 * R.lam) / R.cnt;
    A.rmsd[(size_t)i * A.m + j] = sqrt(ms > 0.0 ? ms : 0.0);
  }
}

// ---- TM-score: one wave per (pair, seed fragment).  The search of the TM-score program as evaluate.tm_score states it: from the
// seed's superposition, up to 20 refinements on the pairs closer than d0_search + 1 (first cut d0_search - 1, cut widened in
// steps of 0.5 while fewer than 3 pairs pass), stop when the set repeats; the score of every superposition counts.
template <int RPL>  // residues per lane: L <= 64 RPL
__global__ __launch_bounds__(256) void k_sup_tm(SupArgs A) {
  const int lane = threadIdx.x & 63;
  // the pair sits on gridDim.x (up to 2^31 - 1 blocks), the seed block on gridDim.y (a few hundred at most): with the pair on
  // y, 257 structures against themselves already exceeded the 65 535 blocks that dimension allows
  const int seed = blockIdx.y * 4 + (threadIdx.x >> 6);
  const long pair = blockIdx.x;
  if (seed >= A.nseed || pair >= A.npair) return;
  int i, j;
  sup_pair_ij(pair, A.n, A.m, A.symmetric, i, j);
  const int L = A.L;
  const float* x = A.xa + (size_t)i * L * 3;
  const float* y = A.xb + (size_t)j * L * 3;
  float px[RPL][3], py[RPL][3];  // the coordinates ARE float32 values (PDB text); the arithmetic below is float64
  bool sel[RPL], ok[RPL];
  const int2 sd = A.seeds[seed];
#pragma unroll
  for (int k = 0; k < RPL; k++) {
    const int r = k * 64 + lane;
    ok[k] = r < L;
    const int rc = ok[k] ? r : 0;
#pragma unroll
    for (int a = 0; a < 3; a++) { px[k][a] = x[rc * 3 + a]; py[k][a] = y[rc * 3 + a]; }
    sel[k] = ok[k] && r >= sd.x && r < sd.x + sd.y;
  }
  const double id02 = 1.0 / (A.d0 * A.d0);
  double best = 0.0;
#pragma unroll 1
  for (int it = 0; it < 21; it++) {
    double v[17];
#pragma unroll
    for (int k = 0; k < 17; k++) v[k] = 0;
#pragma unroll
    for (int k = 0; k < RPL; k++)
      if (sel[k]) {
        v[0] += 1.0;
#pragma unroll
        for (int a = 0; a < 3; a++) {
          v[1 + a] += (double)px[k][a]; v[4 + a] += (double)py[k][a];
#pragma unroll
          for (int b = 0; b < 3; b++) v[7 + a * 3 + b] += (double)px[k][a] * (double)py[k][b];
        }
      }
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = wave_sum(v[k]);  // v[16] (e0) is not needed here
    const Sup R = sup_from_sums(v);
    double d2[RPL], sc = 0.0;
#pragma unroll
    for (int k = 0; k < RPL; k++) {
      double t[3], dd = 0;
#pragma unroll
      for (int a = 0; a < 3; a++) t[a] = (double)px[k][a] - R.cx[a];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const double u = R.M[a][0] * t[0] + R.M[a][1] * t[1] + R.M[a][2] * t[2] + R.cy[a] - (double)py[k][a];
        dd += u * u;
      }
      d2[k] = dd;
      if (ok[k]) sc += 1.0 / (1.0 + dd * id02);
    }
    sc = wave_sum(sc) / A.lnorm;
    best = sc > best ? sc : best;
    double d = it == 0 ? A.d0_search - 1.0 : A.d0_search + 1.0;
    bool nw[RPL];
    while (true) {
      int cnt = 0;
#pragma unroll
      for (int k = 0; k < RPL; k++) { nw[k] = ok[k] && d2[k] < d * d; cnt += __popcll(__ballot(nw[k])); }
      // d2 is finite for finite input (the C ABI rejects anything else); the bound on d keeps a wave from spinning for ever
      // should a non-finite value get here all the same (NaN < d * d is false for every d)
      if (cnt >= 3 || L <= 3 || !(d < 1.0e4)) break;
      d += 0.5;
    }
    bool same = true;
#pragma unroll
    for (int k = 0; k < RPL; k++) same = same && (__ballot(nw[k] != sel[k]) == 0ull);
    if (it > 0 && same) break;
#pragma unroll
    for (int k = 0; k < RPL; k++) sel[k] = nw[k];
  }
  if (lane == 0) atomicMax(A.tm_bits + (size_t)i * A.m + j, (unsigned long long)__double_as_longlong(best));
}
