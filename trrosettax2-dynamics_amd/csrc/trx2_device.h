// trx2_device.h -- small device-side math helpers shared by the trx2fold kernels (gfx950 / wave64).
#pragma once
#include <hip/hip_runtime.h>

#define TRX2_PI_F 3.14159265358979323846f
#define TRX2_DEG_F (TRX2_PI_F / 180.0f)

struct f3 {
  float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3& operator+=(f3& a, f3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
  return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ f3 fma3(f3 a, float s, f3 acc) {  // acc + a*s
  return f3{fmaf(a.x, s, acc.x), fmaf(a.y, s, acc.y), fmaf(a.z, s, acc.z)};
}
__device__ __forceinline__ f3 unit(f3 a) { return a * rsqrtf(dot(a, a)); }

// ---- rigid transform x -> R x + t (row-major R) ----------------------------------------------------------
struct Xf {
  float r[9];
  float t[3];
};
__device__ __forceinline__ Xf xf_identity() { return Xf{{1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 0, 0}}; }
__device__ __forceinline__ f3 xf_apply(const Xf& A, f3 p) {
  return f3{fmaf(A.r[0], p.x, fmaf(A.r[1], p.y, fmaf(A.r[2], p.z, A.t[0]))),
            fmaf(A.r[3], p.x, fmaf(A.r[4], p.y, fmaf(A.r[5], p.z, A.t[1]))),
            fmaf(A.r[6], p.x, fmaf(A.r[7], p.y, fmaf(A.r[8], p.z, A.t[2])))};
}
// (A o B)(x) = A(B(x))
__device__ __forceinline__ Xf xf_compose(const Xf& A, const Xf& B) {
  Xf C;
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++)
      C.r[i * 3 + j] = fmaf(A.r[i * 3], B.r[j], fmaf(A.r[i * 3 + 1], B.r[3 + j], A.r[i * 3 + 2] * B.r[6 + j]));
    C.t[i] = fmaf(A.r[i * 3], B.t[0], fmaf(A.r[i * 3 + 1], B.t[1], fmaf(A.r[i * 3 + 2], B.t[2], A.t[i])));
  }
  return C;
}
// frame with origin at CA, ex along CA->C, ey towards N in the N-CA-C plane, ez = ex x ey
__device__ __forceinline__ Xf xf_from_atoms(f3 N, f3 CA, f3 C) {
  f3 ex = unit(C - CA);
  f3 v = N - CA;
  f3 ey = unit(v - ex * dot(v, ex));
  f3 ez = cross(ex, ey);
  return Xf{{ex.x, ey.x, ez.x, ex.y, ey.y, ez.y, ex.z, ey.z, ez.z}, {CA.x, CA.y, CA.z}};
}
// DPP move of all twelve components: lanes the control gives no source (or whose row the mask leaves out) keep their own value
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Xf xf_dpp(const Xf& v) {
  Xf o;
#pragma unroll
  for (int i = 0; i < 9; i++) o.r[i] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v.r[i]), __float_as_int(v.r[i]), CTRL, ROW_MASK, 0xF, false));
#pragma unroll
  for (int i = 0; i < 3; i++) o.t[i] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v.t[i]), __float_as_int(v.t[i]), CTRL, ROW_MASK, 0xF, false));
  return o;
}
// Inclusive scan of rigid transforms over the 64 lanes of a wave: P_i = M_0 o M_1 o ... o M_i.  Kogge-Stone inside each row of 16
// lanes on DPP row shifts (row_shr:1, 2, 4, 8), then the two row broadcasts of the gfx9 wave scan (row_bcast:15 into rows 1 and 3,
// row_bcast:31 into rows 2 and 3): six rounds of twelve DPP moves at VALU speed.  (__shfl_up is a ds_bpermute through the LDS
// crossbar: twelve of them per round were ~220 of a round's ~370 cycles, profiles/README.md round 3.)
__device__ __forceinline__ Xf xf_wave_scan(Xf P, int lane) {
  const int li = lane & 15;
  { const Xf t = xf_dpp<0x111, 0xF>(P); if (li >= 1) P = xf_compose(t, P); }
  { const Xf t = xf_dpp<0x112, 0xF>(P); if (li >= 2) P = xf_compose(t, P); }
  { const Xf t = xf_dpp<0x114, 0xF>(P); if (li >= 4) P = xf_compose(t, P); }
  { const Xf t = xf_dpp<0x118, 0xF>(P); if (li >= 8) P = xf_compose(t, P); }
  { const Xf t = xf_dpp<0x142, 0xA>(P); if (lane & 16) P = xf_compose(t, P); }
  { const Xf t = xf_dpp<0x143, 0xC>(P); if (lane & 32) P = xf_compose(t, P); }
  return P;
}
// the value of the lane below (lane 0 keeps its own): wave_shr:1
__device__ __forceinline__ Xf xf_from_lane_below(const Xf& v) { return xf_dpp<0x138, 0xF>(v); }
// place atom d: |cd| = len, angle(b,c,d) = ang, dihedral(a,b,c,d) = tor   (cs = cos/sin of ang, tor)
__device__ __forceinline__ f3 place_atom(f3 a, f3 b, f3 c, float len, float cang, float sang, float ctor, float stor) {
  f3 bc = unit(c - b);
  f3 n = unit(cross(b - a, bc));
  f3 m = cross(n, bc);
  return c + bc * (-len * cang) + m * (len * sang * ctor) + n * (len * sang * stor);
}

// Decoy-minor coordinate record of a residue (xyzT, read by the pair kernel): four float4 holding CA N CB C O in THAT order
// -- the contact scan reads the first float4 alone (CA), a restraint visit the first three (CA, N, CB), the contact walk all.
// (The decoy-major record P keeps the order N CA C O CB: it is the Cartesian role's degree-of-freedom vector.)
__device__ __forceinline__ void xt_pack(f3 N, f3 CA, f3 C, f3 O, f3 CB, float4& q0, float4& q1, float4& q2, float4& q3) {
  q0 = make_float4(CA.x, CA.y, CA.z, N.x);
  q1 = make_float4(N.y, N.z, CB.x, CB.y);
  q2 = make_float4(CB.z, C.x, C.y, C.z);
  q3 = make_float4(O.x, O.y, O.z, 0.0f);
}
__device__ __forceinline__ void xt_unpack(float4 q0, float4 q1, float4 q2, float4 q3, f3& CA, f3& N, f3& CB, f3& C, f3& O) {
  CA = mk3(q0.x, q0.y, q0.z); N = mk3(q0.w, q1.x, q1.y); CB = mk3(q1.z, q1.w, q2.x); C = mk3(q2.y, q2.z, q2.w); O = mk3(q3.x, q3.y, q3.z);
}
// the same from a decoy-major record (N CA C O CB, 15 floats + pad)
__device__ __forceinline__ void xt_pack_from_p(float4 p0, float4 p1, float4 p2, float4 p3, float4& q0, float4& q1, float4& q2, float4& q3) {
  xt_pack(mk3(p0.x, p0.y, p0.z), mk3(p0.w, p1.x, p1.y), mk3(p1.z, p1.w, p2.x), mk3(p2.y, p2.z, p2.w), mk3(p3.x, p3.y, p3.z), q0, q1, q2, q3);
}

// hardware reciprocal (v_rcp_f32, 1 ulp) for the geometric factors below: an IEEE division is ~10 vector instructions and
// the pair kernel, which is vector-ALU-bound (profiles/README.md), evaluates up to eight of them per residue-pair visit
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
// 1/sqrt and sqrt straight from the hardware instruction (1 ulp): rsqrtf() / sqrtf() wrap it in a rescaling for denormal arguments
// (five more instructions each), and squared interatomic distances are never denormal
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// atan2 for the restraint angles: the library's is 43 vector instructions, this is ~23.  atan(a) = a P(a^2) on [0,1] (degree-7
// fit, max error 1.4e-7 rad evaluated in float32 = 1-2 ulp at pi/4), octant fix-ups by selects.  atan2(0,0) = 0 as in libm.
__device__ __forceinline__ float fast_atan2f(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y), mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float a = mn * frcp(fmaxf(mx, 1e-30f)), s = a * a;
  float p = -4.054523205e-03f;
  p = fmaf(p, s, 2.186279171e-02f);
  p = fmaf(p, s, -5.591207582e-02f);
  p = fmaf(p, s, 9.642178046e-02f);
  p = fmaf(p, s, -1.390862164e-01f);
  p = fmaf(p, s, 1.994656400e-01f);
  p = fmaf(p, s, -3.332986064e-01f);
  p = fmaf(p, s, 9.999993355e-01f);
  float r = p * a;
  r = ay > ax ? 0.5f * TRX2_PI_F - r : r;
  r = x < 0.0f ? TRX2_PI_F - r : r;
  return y < 0.0f ? -r : r;
}
// sin and cos for torsion / bond angles: the library's sincosf is 122 vector instructions (it carries the Payne-Hanek path for
// huge arguments); the step kernel needs ten per residue and step.  Cody-Waite reduction to [-pi/4, pi/4] with pi/2 split in
// two floats (the fused multiply-adds keep the products exact), cephes polynomials; max error 9e-8 for |x| < 1000, checked
// in float32 arithmetic against double.  Torsions are bounded by the line search to a few turns.
__device__ __forceinline__ void fast_sincosf(float x, float* sn, float* cs) {
  const float k = rintf(x * 0.63661977236758134f);
  float r = fmaf(k, -1.5707963705062866f, x);
  r = fmaf(k, 4.3711390063e-08f, r);
  const float s2 = r * r;
  const float sp = fmaf(fmaf(-1.9515295891e-4f, s2, 8.3321608736e-3f), s2, -1.6666654611e-1f);
  const float s_ = fmaf(r * s2, sp, r);
  const float cp = fmaf(fmaf(2.443315711809948e-5f, s2, -1.388731625493765e-3f), s2, 4.166664568298827e-2f);
  const float c_ = fmaf(s2 * s2, cp, fmaf(-0.5f, s2, 1.0f));
  const int q = (int)k;
  const float s1 = (q & 1) ? c_ : s_, c1 = (q & 1) ? s_ : c_;
  *sn = (q & 2) ? -s1 : s1;
  *cs = ((q + 1) & 2) ? -c1 : c1;
}

// IUPAC dihedral p1-p2-p3-p4 and its gradient with respect to the four points
__device__ __forceinline__ float dihedral_grad(f3 p1, f3 p2, f3 p3, f3 p4, f3& d1, f3& d2, f3& d3, f3& d4) {
  f3 F = p1 - p2, G = p2 - p3, H = p4 - p3;
  f3 A = cross(F, G), B = cross(H, G);
  float G2 = dot(G, G);
  float iGn = frsq(G2), Gn = G2 * iGn;
  float iA2 = frcp(fmaxf(dot(A, A), 1e-12f)), iB2 = frcp(fmaxf(dot(B, B), 1e-12f));
  float cosv = dot(A, B), sinv = dot(cross(B, A), G) * iGn;
  float ang = fast_atan2f(sinv, cosv);
  float ca = dot(F, G) * iA2 * iGn, cb = dot(H, G) * iB2 * iGn;
  float ga = Gn * iA2, gb = Gn * iB2;
  d1 = A * (-ga);
  d4 = B * gb;
  d2 = A * (ga + ca) - B * cb;
  d3 = B * (cb - gb) - A * ca;
  return ang;
}

// the value alone (same operations as dihedral_grad's)
__device__ __forceinline__ float dihedral_val(f3 p1, f3 p2, f3 p3, f3 p4) {
  f3 F = p1 - p2, G = p2 - p3, H = p4 - p3;
  f3 A = cross(F, G), B = cross(H, G);
  float iGn = frsq(dot(G, G));
  return fast_atan2f(dot(cross(B, A), G) * iGn, dot(A, B));
}

// planar angle p1-p2-p3 in [0,pi] and gradient
__device__ __forceinline__ float angle_grad(f3 p1, f3 p2, f3 p3, f3& d1, f3& d2, f3& d3) {
  f3 v = p1 - p2, w = p3 - p2;
  float ivn = frsq(dot(v, v)), iwn = frsq(dot(w, w));
  f3 vh = v * ivn, wh = w * iwn;
  float c = fminf(1.0f, fmaxf(-1.0f, dot(vh, wh)));
  const float sn = fsqrt(1.0f - c * c);  // sin of the angle (>= 0): the angle is atan2(sin, cos) by the same polynomial as the dihedrals
  float ang = fast_atan2f(sn, c);     // (acosf() is ~35 instructions)
  float is = -frcp(fmaxf(sn, 1e-8f));
  d1 = (wh - vh * c) * (is * ivn);
  d3 = (vh - wh * c) * (is * iwn);
  d2 = (d1 + d3) * -1.0f;
  return ang;
}

// LDS hand-off between the lanes of ONE wave: its LDS instructions execute in program order, so only the compiler has to be
// kept from moving them across this point
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---- wave-wide sum, total in every lane; all 64 lanes must be active.
// __shfl_xor is a ds_bpermute through the LDS crossbar (~120 cycles, two per double): six dependent rounds made one
// reduction ~1000+ cycles, and a minimiser step has ~27 of them (s_memtime stamps, profiles/README.md).  DPP row operations
// do the butterfly inside each row of 16 lanes in a few cycles per round; v_readlane then combines the four rows.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_value(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ double lane_value(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// A value that IS the same in every lane, marked so for the compiler (v_readfirstlane: the result lives in scalar registers and
// every branch on it is a scalar branch).  Used on the step kernels' energy totals: they come out of LDS reductions and are
// wave-uniform by construction, but a value the compiler cannot PROVE uniform makes it if-convert the minimiser's state machine
// under exec masks -- and in one such build (one-sum energy path with two residues per thread, kernel_step.h) the masked arm
// of the pre-checked start lost its `fh[0] = f` assignment ("the one-sum path").
__device__ __forceinline__ double uniform_d(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <typename T>
__device__ __forceinline__ T wave_sum_dpp(T v) {
  v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]: lane ^ 2
  v += dpp_move<0x141>(v);  // row_half_mirror: lane i <- 7 - i of its group of 8 (the other quad)
  v += dpp_move<0x140>(v);  // row_mirror: lane i <- 15 - i of its row (the other half)
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp(v); }
__device__ __forceinline__ double wave_sum(double v) { return wave_sum_dpp(v); }
// Inclusive SUFFIX sums over the wave (lane i: sum over lanes >= i) of K floats: row_shl:1, 2, 4, 8 inside the rows of 16, then
// the totals of the rows above (lanes 16, 32, 48 after the row stage) by v_readlane.
template <int K>
__device__ __forceinline__ void wave_suffix_sums(float (&v)[K], int lane) {
  const int li = lane & 15;
#pragma unroll
  for (int i = 0; i < K; i++) {
    float x = v[i], t;
    t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x101, 0xF, 0xF, false)); if (li < 15) x += t;
    t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x102, 0xF, 0xF, false)); if (li < 14) x += t;
    t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x104, 0xF, 0xF, false)); if (li < 12) x += t;
    t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x108, 0xF, 0xF, false)); if (li < 8) x += t;
    const float r1 = lane_value(x, 16), r2 = lane_value(x, 32), r3 = lane_value(x, 48);
    const float above = lane < 16 ? (r1 + (r2 + r3)) : lane < 32 ? (r2 + r3) : lane < 48 ? r3 : 0.0f;
    v[i] = x + above;
  }
}

