// trx2fold.hip -- MI355X (gfx950, wave64) kernels + C ABI of the restraint-guided batched fold.
//
// Replaces the per-decoy PyRosetta process of /root/reference/folding/folding.py (+ folding/utils_ros) by a
// device-resident batched minimiser.  Kernel inventory ("native work-list"):
//   K2 k_build_tables : gen_rst + add_rst selection -> dense (y, y'') spline tables  (utils_ros.py:6-146,706-723)
//   K3/K4 k_pair<BW>  : CB-CB distance / omega / theta / phi spline restraints + soft-sphere repulsion + backbone hydrogen
//                       bonds, energy and Cartesian gradient, lane = decoy                 (folding.py:74-84 score terms)
//   K1/K5/K6 k_chain  : per decoy: gradient slabs -> torsion gradient (suffix scan of force/torque), rama/omega
//                       terms, non-monotone Armijo L-BFGS state machine over the staged protocol, new trial
//                       torsions -> backbone by a parallel rigid-transform scan (NeRF) (folding.py:86-119,164-171)
// Data layout in HBM: see DESIGN.md.  No CPU fallback exists: every entry point runs on the GPU or fails.
// One translation unit: kernel_tables.h (K2), kernel_pair.h (K3/K4), kernel_step.h (K1/K5/K6 + Cartesian role) are
// included below the shared macros and constant tables; this file holds those and the host side (context, C ABI).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <condition_variable>
#include <functional>
#include <mutex>
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
#define RED_STRIDE 25

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

#include "kernel_tables.h"
#include "kernel_pair.h"
#include "kernel_step.h"
#include "kernel_feedback.h"
#include "kernel_superpose.h"

// =================================================================================================
// host side
// =================================================================================================
#define TRX2_NS_CAP 8  /* slices per row at most (4 bits in a work item) */
struct RowPlan {
  int pw = 0, groups = 0;       // launch shape: partner residues per wave step (64 / decoys per wave), decoy groups
  long epoch = -1;              // rows_epoch the plan was built for
  int forced = 0;               // TRX2_NSPLIT it was built under (0: the rule)
  int n_items = 0, ns_max = 1;
  double ns_avg = 1;
  uint2* items = nullptr;           // device [n_items]
  unsigned char* nslice = nullptr;  // device [L]
  size_t cap_items = 0, cap_L = 0;
};
// Host thread of a context's second lane (trx2_ctx_set_lanes): started with the lane, parked on a condition variable between
// calls, joined when the lane goes -- a two-lane call hands it one job instead of creating a thread of its own every time.
struct LaneWorker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, done = false, quit = false;
  int rc = 0;
  void main_loop() {
    std::unique_lock<std::mutex> lk(mu);
    while (true) {
      cv.wait(lk, [&] { return has_job || quit; });
      if (quit) return;
      std::function<int()> f = std::move(job);
      has_job = false;
      lk.unlock();
      const int r = f();
      lk.lock();
      rc = r; done = true;
      cv.notify_all();
    }
  }
  void start() { th = std::thread([this] { main_loop(); }); }
  void submit(std::function<int()> f) { std::lock_guard<std::mutex> lk(mu); job = std::move(f); has_job = true; done = false; cv.notify_all(); }
  int wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done; }); done = false; return rc; }
  void stop() {
    { std::lock_guard<std::mutex> lk(mu); quit = true; cv.notify_all(); }
    if (th.joinable()) th.join();
  }
};

struct trx2_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // map
  int L = 0, use_orient = 0;
  std::string seq;
  float2 *Td = nullptr, *To = nullptr, *Tt = nullptr, *Tp = nullptr;
  float *pd = nullptr, *po = nullptr, *pt = nullptr, *pp = nullptr;
  unsigned char *gen = nullptr, *sel = nullptr, *mask2 = nullptr, *hasH = nullptr;
  // restraint variants: idr = pair flags of gen_idp_rst / mode 3 (NULL: none); mask_odr = packed masks without
  // the flagged pairs; rst_kind 0 gen_rst, 1 gen_idp_rst, 2 gen_rst_af2 (kd = 60 knots, distance on C-alpha, 64-bin map)
  unsigned char *idr = nullptr, *mask_odr = nullptr;
  double* idr_bk = nullptr;
  // row lists of the pair kernel (k_build_rows): row a's partners with any selected restraint, [L][L] entries + [L] counts
  unsigned* rows = nullptr; int* row_cnt = nullptr;
  // relax stage (a11-lite): packed masks of the two re-selections, the entries' copies of them, glycine flags
  unsigned char *mask_r1 = nullptr, *mask_r2 = nullptr, *gly = nullptr; unsigned short* rows_rx = nullptr;
  int rst_kind = 0, kd = KD, dist_ca = 0;
  double knots_af2_last = 0;
  float* knots_f = nullptr;
  double* knots_d = nullptr;
  double knots_h[TRX2_KTOT_MAX];
  // batch
  int Bcap = 0, Lcap = 0, BW = 64, Bpad = 0;
  size_t fa_cap = 0;     // pair-kernel records the FA buffer holds: (slices x slots) units of L records
  // Row plans of the pair kernel: how many slices (workgroups) each row of the restraint lists is cut into for a launch shape
  // (decoys per wave, decoy groups); built on the host from the rows' list lengths, cached per shape until the lists change.
  std::vector<int> h_row_cnt;    // host copy of row_cnt (build_tables)
  long rows_epoch = 0;           // bumped whenever the lists are rebuilt
  std::vector<RowPlan> plans;
  int plan_cur = -1;             // index into plans: the shape the next launches use
  int* plan = nullptr;   // compaction plan (device)
  int compact = 1;       // trx2_ctx_set_tail_compaction
  int* st_i = nullptr; double* st_d = nullptr; float* rho = nullptr; double* gram = nullptr;
  float4 *X = nullptr, *G = nullptr, *D = nullptr, *XT = nullptr, *S = nullptr, *Y = nullptr;
  float4* P = nullptr; float4* xyzT = nullptr; float* wcur = nullptr; float4* geom = nullptr;
  float* FA = nullptr;
  // segment cache of the pair kernel (kernel_pair.h, PairArgs): blocks + tags for Bpad x L x L (entry, decoy) places; tab_epoch counts
  // table changes, seg_epoch is the one the tags were last reset for
  float4* segc = nullptr; uint2* segt = nullptr; size_t seg_bytes = 0, seg_elems = 0; long tab_epoch = 0, seg_epoch = -1;
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
  hipEvent_t ev_ready = nullptr;  // shared launches: recorded on this context's stream when a fold's start-up work is enqueued
  double last_seconds = 0; int last_launches = 0; double last_slot_eff = 0;
  // slot pool (trx2_ctx_set_pool): a fold of N decoys runs on min(N, pool) slots; a slot whose decoy has reported takes the next
  // one of the queue on the device.  0 = one slot per decoy.
  int pool = 0;
  // single-decoy folds: waves per row of the pair kernel -- 4 (default: the shortest evaluation when few folds are in flight) or 1
  // (k_pair1: a quarter of the waves, the shape for MANY concurrent single-decoy folds sharing launches); trx2_ctx_set_single_decoy_waves
  int pair1_waves = 4;
  int *slot_id = nullptr, *next_id = nullptr, *out_stat = nullptr;
  float4 *out_xyz = nullptr, *out_X = nullptr; double *out_e = nullptr, *out_f = nullptr; float* tors0_all = nullptr;
  size_t out_cap = 0, out_L = 0;  // decoys x residues the output arrays hold
  // trx2_ctx_set_profiling: every prof_every-th evaluation of a fold is bracketed by HIP events on the stream (pair | step)
  int prof_every = 0;
  int step_dyn_max[2] = {0, 0};  // dynamic LDS the fused step kernels may ask for (L <= 128 | L <= 256): 160 KB - their static LDS
  int step_dyn_floor[2] = {0, 0};  // ... and the most the torsion role needs of it (its staged history at 128 / 256 residues)
  int step_static[2] = {0, 0};     // static LDS of the two fused step kernels
  int step2_static[2] = {0, 0};    // ... and of their low-register instantiations (k_step<1, NT, NT, true>)
  int half_static[2] = {0, 0};     // ... and of the half-evaluation kernels of the shared launches (k_half_multi<.., 128 | 256>)
  int pair_static = 32 * 1024;     // static LDS of a pair-kernel workgroup
  int lds_total = 160 * 1024;      // LDS of a CU
  std::vector<hipEvent_t> prof_ev;
  double prof_pair_ms = 0, prof_step_ms = 0; int prof_n = 0;
  std::vector<float> prof_a, prof_b;   // the samples themselves: the accessor drops those beyond four medians (a host stall between an event and its launch)
  // second lane (trx2_ctx_set_lanes): a context of its own stream and batch buffers that BORROWS this one's tables, so that
  // one job can run as two half-batches whose pair and step kernels overlap
  trx2_ctx* child = nullptr;
  LaneWorker* lane_worker = nullptr;   // the child's host thread (lives as long as the child)
  bool borrows_map = false;
  // feedback scratch (trx2_feedback_*): grows on demand
  void* fb_buf = nullptr; size_t fb_cap = 0;
  // the current distograms stay resident (dist, omega, theta, phi) so that the feedback step can re-weight them in place and
  // rebuild the tables without a host round trip; alt = output buffers of that step, tmp = the cumulative convergence array
  float* cur[4] = {nullptr, nullptr, nullptr, nullptr};
  float* alt[4] = {nullptr, nullptr, nullptr, nullptr};
  float *tmp_cur = nullptr, *tmp_alt = nullptr;
  bool has_tmp = false;
  trx2_params prm;
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

static std::atomic<int> g_lane_contexts{0};  // ... of which second lanes (trx2_ctx_set_lanes)
static std::atomic<int> g_live_contexts{0};  // contexts alive in this process (lanes are contexts too): step_dyn_budget

// Stream pool.  HIP maps a process's streams onto GPU_MAX_HW_QUEUES = 4 hardware queues, and a stream that has launched work keeps
// its queue while it lives: with two used streams left alive (contexts a caller cached and never closed: bench.py's e2e leg did),
// the streams of the next two contexts came to share ONE hardware queue -- their kernels serialised and two concurrent chains
// folded at 493 instead of 805 decoys/s (profiles/README.md, round 3).  So the library owns four streams per device, created
// together (consecutive hardware queues), for the life of the process, and a context takes the least-used one: up to four
// concurrently folding contexts (two chains x two lanes) never share a queue, whatever was created, used and leaked before.
// More than four share streams in turn (their work serialises per stream, which two streams on one queue did anyway).
#include <map>
#include <mutex>
#define TRX2_POOL_STREAMS 4
struct StreamPool { hipStream_t s[TRX2_POOL_STREAMS] = {nullptr, nullptr, nullptr, nullptr}; int use[TRX2_POOL_STREAMS] = {0, 0, 0, 0}; bool ready = false; };
static std::mutex g_pool_mutex;
static std::map<int, StreamPool> g_pools;
// avoid: a stream the caller must not share (a second lane takes any stream but its parent's: after contexts have come and gone
// the least-used stream can be the parent's own, and two lanes on one stream serialise -- ADVICE r3)
static hipStream_t pool_acquire(int device, hipStream_t avoid = nullptr, int weight = 1) {   // weight: a launch engine's stream counts as heavily used
  std::lock_guard<std::mutex> lk(g_pool_mutex);
  StreamPool& P = g_pools[device];
  if (!P.ready) {
    for (int k = 0; k < TRX2_POOL_STREAMS; k++)
      if (hipStreamCreateWithFlags(&P.s[k], hipStreamNonBlocking) != hipSuccess) {
        for (int q = 0; q < k; q++) { (void)hipStreamDestroy(P.s[q]); P.s[q] = nullptr; }
        return nullptr;
      }
    P.ready = true;
  }
  int best = -1;
  for (int k = 0; k < TRX2_POOL_STREAMS; k++) if (P.s[k] != avoid && (best < 0 || P.use[k] < P.use[best])) best = k;
  P.use[best] += weight;
  return P.s[best];
}
// contexts that hold this stream (more than one: their launches interleave on it -- no stream capture then)
static int pool_use_count(int device, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_pool_mutex);
  auto it = g_pools.find(device);
  if (it == g_pools.end()) return 0;
  for (int k = 0; k < TRX2_POOL_STREAMS; k++) if (it->second.s[k] == st) return it->second.use[k];
  return 0;
}
static void pool_release(int device, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_pool_mutex);
  auto it = g_pools.find(device);
  if (it == g_pools.end()) return;
  for (int k = 0; k < TRX2_POOL_STREAMS; k++) if (it->second.s[k] == st && it->second.use[k] > 0) it->second.use[k]--;
}

extern "C" int trx2_abi_version(void) { return 2; }

static int ctx_create_impl(int device, trx2_ctx** out, hipStream_t avoid_stream);
extern "C" int trx2_ctx_create(int device, trx2_ctx** out) { return ctx_create_impl(device, out, nullptr); }
static int ctx_create_impl(int device, trx2_ctx** out, hipStream_t avoid_stream) {
  if (!out) return 1;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return 2;
  trx2_ctx* ctx = new trx2_ctx();
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess || (ctx->stream = pool_acquire(device, avoid_stream)) == nullptr) {
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
  // The step kernels of short chains stage the L-BFGS history in dynamic LDS: the torsion role all of it (HIST_LDS_BYTES(L): 8
  // pairs in rows L long, 38 KB at 150 residues, 64 KB at 256), the Cartesian role as many pairs (CART_HIST_BYTES(L) each) as fit
  // beside the kernel's static LDS in the 160 KB of a gfx950 workgroup.
  {
    int lds_max = 0;
    if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess || lds_max <= 0) lds_max = 65536;
    if (lds_max > 160 * 1024) lds_max = 160 * 1024;
    const void* fstep[2] = {(const void*)k_step<1, 128, 128>, (const void*)k_step<1, CHAIN_THREADS, CHAIN_THREADS>};
    const int hist[2] = {HIST_LDS_BYTES(128), HIST_LDS_BYTES(CHAIN_THREADS)};
    bool ok = hipFuncSetAttribute((const void*)k_chain<1, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, hist[0]) == hipSuccess &&
              hipFuncSetAttribute((const void*)k_chain<1, CHAIN_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, hist[1]) == hipSuccess;
    {  // 256 < L <= 512: the torsion history (256 L bytes) in dynamic LDS too
      const void* f512[2] = {(const void*)k_chain<1, 2 * CHAIN_THREADS>, (const void*)k_step<1, 2 * CHAIN_THREADS, 2 * CHAIN_THREADS>};
      for (int k = 0; k < 2 && ok; k++) {
        hipFuncAttributes fa;
        ok = hipFuncGetAttributes(&fa, f512[k]) == hipSuccess && lds_max - (int)fa.sharedSizeBytes >= (int)HIST_LDS_BYTES(2 * CHAIN_THREADS) &&
             hipFuncSetAttribute(f512[k], hipFuncAttributeMaxDynamicSharedMemorySize, lds_max - (int)fa.sharedSizeBytes) == hipSuccess;
      }
    }
    for (int k = 0; k < 2 && ok; k++) {
      hipFuncAttributes fa;
      ok = hipFuncGetAttributes(&fa, fstep[k]) == hipSuccess;
      if (!ok) break;
      int dyn = lds_max - (int)fa.sharedSizeBytes;  // what a launch may ask for at most; how much it does ask for: step_dyn_budget()
      if (dyn < hist[k]) dyn = hist[k];
      ctx->step_dyn_max[k] = dyn;
      ctx->step_dyn_floor[k] = hist[k];
      ctx->step_static[k] = (int)fa.sharedSizeBytes;
      ok = hipFuncSetAttribute(fstep[k], hipFuncAttributeMaxDynamicSharedMemorySize, dyn) == hipSuccess;
    }
    if (ok) {
      hipFuncAttributes fp;
      ok = hipFuncGetAttributes(&fp, (const void*)k_pair<64, FAM_ALL>) == hipSuccess;
      if (ok) ctx->pair_static = (int)fp.sharedSizeBytes;
    }
    // the low-register instantiations (folds on many slots): two workgroups of 256 threads per CU, four of 128
    {
      const void* f2[2] = {(const void*)k_step<1, 128, 128, true>, (const void*)k_step<1, CHAIN_THREADS, CHAIN_THREADS, true>};
      ctx->lds_total = lds_max;
      for (int k = 0; k < 2 && ok; k++) {
        hipFuncAttributes fa;
        ok = hipFuncGetAttributes(&fa, f2[k]) == hipSuccess;
        if (!ok) break;
        ctx->step2_static[k] = (int)fa.sharedSizeBytes;
        ok = hipFuncSetAttribute(f2[k], hipFuncAttributeMaxDynamicSharedMemorySize, lds_max - (int)fa.sharedSizeBytes) == hipSuccess;
      }
    }
    {  // the half-evaluation kernels of the shared launches (kernel_step.h, k_half_multi): same occupancy as the low-register step kernels
      const void* fh[2][2] = {{(const void*)k_half_multi<FAM_ALL, true, 128>, (const void*)k_half_multi<FAM_DIST | FAM_VDW, true, 128>},
                              {(const void*)k_half_multi<FAM_ALL, true, CHAIN_THREADS>, (const void*)k_half_multi<FAM_DIST | FAM_VDW, true, CHAIN_THREADS>}};
      for (int k = 0; k < 2 && ok; k++)
        for (int q = 0; q < 2 && ok; q++) {
          hipFuncAttributes fa;
          ok = hipFuncGetAttributes(&fa, fh[k][q]) == hipSuccess;
          if (!ok) break;
          ctx->half_static[k] = std::max(ctx->half_static[k], (int)fa.sharedSizeBytes);
          ok = hipFuncSetAttribute(fh[k][q], hipFuncAttributeMaxDynamicSharedMemorySize, lds_max - (int)fa.sharedSizeBytes) == hipSuccess;
        }
    }
    if (!ok) {
      pool_release(device, ctx->stream);
      delete ctx;
      return 4;
    }
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
    pool_release(device, ctx->stream);
    delete ctx;
    return 4;
  }
  g_live_contexts++;
  *out = ctx;
  return 0;
}

static void lend_map(trx2_ctx* c);
static void free_plans(trx2_ctx* c) {
  for (auto& p : c->plans) { if (p.items) (void)hipFree(p.items); if (p.nslice) (void)hipFree(p.nslice); }
  c->plans.clear(); c->plan_cur = -1;
}
static void free_map(trx2_ctx* c) {
  free_plans(c);
  if (c->child) free_plans(c->child);
  if (c->child) {  // the borrower must be idle and forget the tables before they go
    if (c->child->stream) (void)hipStreamSynchronize(c->child->stream);
    trx2_ctx* k = c->child;
    k->Td = k->To = k->Tt = k->Tp = nullptr; k->pd = k->po = k->pt = k->pp = nullptr;
    k->gen = k->sel = k->mask2 = k->hasH = k->idr = k->mask_odr = nullptr; k->knots_f = nullptr; k->knots_d = nullptr; k->L = 0; k->alloc_epoch++;
    k->rows = nullptr; k->row_cnt = nullptr; k->h_row_cnt.clear(); k->rows_epoch++;
    k->mask_r1 = k->mask_r2 = k->gly = nullptr; k->rows_rx = nullptr;
  }
  void* p[] = {c->Td, c->To, c->Tt, c->Tp, c->pd, c->po, c->pt, c->pp, c->gen, c->sel, c->mask2, c->hasH, c->knots_f, c->knots_d,
               c->idr, c->mask_odr, c->idr_bk, c->rows, c->row_cnt, c->mask_r1, c->mask_r2, c->gly, c->rows_rx,
               c->cur[0], c->cur[1], c->cur[2], c->cur[3], c->alt[0], c->alt[1], c->alt[2], c->alt[3], c->tmp_cur, c->tmp_alt};
  for (void* q : p)
    if (q && !c->borrows_map) (void)hipFree(q);
  for (int k = 0; k < 4; k++) c->cur[k] = c->alt[k] = nullptr;
  c->tmp_cur = c->tmp_alt = nullptr; c->has_tmp = false;
  c->Td = c->To = c->Tt = c->Tp = nullptr;
  c->pd = c->po = c->pt = c->pp = nullptr;
  c->gen = c->sel = c->mask2 = c->hasH = c->idr = c->mask_odr = nullptr; c->idr_bk = nullptr;
  c->rows = nullptr; c->row_cnt = nullptr;
  c->mask_r1 = c->mask_r2 = c->gly = nullptr; c->rows_rx = nullptr;
  c->knots_f = nullptr; c->knots_d = nullptr;
  c->rst_kind = 0; c->kd = KD; c->dist_ca = 0;
  c->L = 0;
}
static void free_batch(trx2_ctx* c) {
  void* p[] = {c->st_i, c->st_d, c->rho, c->gram, c->X, c->G, c->D, c->XT, c->S, c->Y, c->P, c->xyzT, c->geom, c->wcur, c->FA,
               c->e_last, c->f_last, c->grad, c->tors0, c->done_count, c->seq_ctr, c->runs, c->plan};
  for (void* q : p)
    if (q) (void)hipFree(q);
  c->st_i = nullptr; c->st_d = nullptr; c->rho = nullptr; c->gram = nullptr;
  c->X = c->G = c->D = c->XT = c->S = c->Y = nullptr;
  c->P = nullptr; c->xyzT = nullptr; c->geom = nullptr; c->wcur = nullptr; c->FA = nullptr;
  c->e_last = c->f_last = nullptr; c->grad = nullptr; c->tors0 = nullptr; c->done_count = nullptr; c->seq_ctr = nullptr; c->runs = nullptr;
  c->plan = nullptr; c->fa_cap = 0;
  if (c->segc) (void)hipFree(c->segc);
  if (c->segt) (void)hipFree(c->segt);
  c->segc = nullptr; c->segt = nullptr; c->seg_bytes = c->seg_elems = 0; c->seg_epoch = -1;
  c->Bcap = c->Lcap = 0;
  c->alloc_epoch++;
  void* q[] = {c->CX, c->CG, c->CD, c->CS, c->CY, c->slot_id, c->next_id, c->out_stat, c->out_xyz, c->out_X, c->out_e, c->out_f, c->tors0_all};
  for (void* v : q)
    if (v) (void)hipFree(v);
  c->CX = c->CG = c->CD = c->CS = c->CY = nullptr;
  c->slot_id = c->next_id = c->out_stat = nullptr; c->out_xyz = c->out_X = nullptr; c->out_e = c->out_f = nullptr; c->tors0_all = nullptr;
  c->out_cap = c->out_L = 0;
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
  k->gen = c->gen; k->sel = c->sel; k->mask2 = c->mask2; k->hasH = c->hasH; k->knots_f = c->knots_f; k->knots_d = c->knots_d;
  k->idr = c->idr; k->mask_odr = c->mask_odr; k->rst_kind = c->rst_kind; k->kd = c->kd; k->dist_ca = c->dist_ca;
  k->rows = c->rows; k->row_cnt = c->row_cnt; k->h_row_cnt = c->h_row_cnt; k->rows_epoch++; k->tab_epoch++;
  k->mask_r1 = c->mask_r1; k->mask_r2 = c->mask_r2; k->gly = c->gly; k->rows_rx = c->rows_rx;
  memcpy(k->knots_h, c->knots_h, sizeof c->knots_h);
  k->alloc_epoch++;
}

extern "C" int trx2_ctx_set_lanes(trx2_ctx* ctx, int lanes) {
  if (!ctx) return 1;
  if (lanes != 1 && lanes != 2) { ctx->err = "trx2_ctx_set_lanes: 1 or 2"; return 1; }
  if (lanes == 2 && !ctx->child) {
    trx2_ctx* k = nullptr;
    if (ctx_create_impl(ctx->device, &k, ctx->stream) != 0) { ctx->err = "trx2_ctx_set_lanes: cannot create the second lane"; return 1; }
    k->borrows_map = true;
    g_lane_contexts++;
    k->pool = ctx->pool;
    k->pair1_waves = ctx->pair1_waves;
    k->compact = ctx->compact;
    ctx->child = k;
    ctx->lane_worker = new LaneWorker();
    ctx->lane_worker->start();
    if (ctx->L) { HIPCHK(hipStreamSynchronize(ctx->stream)); lend_map(ctx); }
  } else if (lanes == 1 && ctx->child) {
    trx2_ctx* k = ctx->child;
    ctx->child = nullptr;
    if (ctx->lane_worker) { ctx->lane_worker->stop(); delete ctx->lane_worker; ctx->lane_worker = nullptr; }
    trx2_ctx_destroy(k);  // borrows_map: frees its batch buffers only
  }
  return 0;
}

extern "C" void trx2_ctx_destroy(trx2_ctx* ctx) {
  if (!ctx) return;
  g_live_contexts--;
  if (ctx->borrows_map) g_lane_contexts--;
  (void)hipSetDevice(ctx->device);
  if (ctx->lane_worker) { ctx->lane_worker->stop(); delete ctx->lane_worker; ctx->lane_worker = nullptr; }
  if (ctx->child) { trx2_ctx* k = ctx->child; ctx->child = nullptr; trx2_ctx_destroy(k); }
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->gexec) (void)hipGraphExecDestroy(ctx->gexec);
  for (auto& e : ctx->prof_ev) (void)hipEventDestroy(e);
  if (ctx->ev_ready) (void)hipEventDestroy(ctx->ev_ready);
  free_map(ctx);
  free_batch(ctx);
  if (ctx->fb_buf) (void)hipFree(ctx->fb_buf);
  if (ctx->h_done) (void)hipHostFree(ctx->h_done);
  if (ctx->stream) pool_release(ctx->device, ctx->stream);  // the stream itself belongs to the pool (synchronised above)
  delete ctx;
}

extern "C" const char* trx2_last_error(const trx2_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

// the rows' list lengths on the host (the row plans are made from them); ends the table build: the stream is idle afterwards
static int fetch_row_counts(trx2_ctx* ctx) {
  ctx->h_row_cnt.assign((size_t)ctx->L, 0);
  HIPCHK(hipMemcpyAsync(ctx->h_row_cnt.data(), ctx->row_cnt, sizeof(int) * (size_t)ctx->L, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->rows_epoch++; ctx->tab_epoch++;
  if (ctx->child) { ctx->child->h_row_cnt = ctx->h_row_cnt; ctx->child->rows_epoch++; ctx->child->tab_epoch++; }
  return 0;
}

// Row plan for a launch shape.  A workgroup's restraint work is its slice of a row's list; rows hold 20 .. 140 partners, so with
// one slice count for every row the launch lasted as long as its longest rows (2 x the mean).  Here a row is cut into
// round(length / target) slices, target = ~18 entries per partner residue a wave step covers (4-5 visits per wave); the target
// shrinks while the launch would have fewer than ~450 workgroups (one resident round is 768 at three waves per SIMD) and grows
// while it would have more than ~2000.  Every wave keeps at least two residues of the contact scan.  Items are ordered longest
// slice first.  TRX2_NSPLIT (A/B timing, and tests that pin the split) forces one slice count for every row;
// TRX2_ROW_TARGET overrides the target.  Deterministic: the same lists and shape give the same plan.
static int plan_get(trx2_ctx* ctx, int pw, int groups, int* index) {
  const int L = ctx->L;
  if ((int)ctx->h_row_cnt.size() != L) { ctx->err = "internal: row counts missing (no map set?)"; return 1; }
  int forced = 0;
  if (const char* e = getenv("TRX2_NSPLIT")) { int v = atoi(e); if (v >= 1 && v <= TRX2_NS_CAP) forced = v; }
  int k = -1;
  for (size_t i = 0; i < ctx->plans.size(); i++)
    if (ctx->plans[i].pw == pw && ctx->plans[i].groups == groups) k = (int)i;
  if (k >= 0 && ctx->plans[(size_t)k].epoch == ctx->rows_epoch && ctx->plans[(size_t)k].forced == forced) { *index = k; return 0; }
  if (k < 0) { ctx->plans.emplace_back(); k = (int)ctx->plans.size() - 1; ctx->plans[(size_t)k].pw = pw; ctx->plans[(size_t)k].groups = groups; }
  RowPlan& P = ctx->plans[(size_t)k];
  const int ns_lim = std::max(1, std::min(TRX2_NS_CAP, L / (2 * PAIR_WAVES * pw)));
  double target = 18.0;
  if (const char* e = getenv("TRX2_ROW_TARGET")) { double v = atof(e); if (v >= 1.0 && v <= 1024.0) target = v; }
  target *= pw;
  std::vector<unsigned char> ns((size_t)L);
  auto make = [&](double t) {
    long tot = 0;
    for (int a = 0; a < L; a++) {
      int n = forced ? forced : (int)std::lround((double)ctx->h_row_cnt[(size_t)a] / t);
      n = std::max(1, std::min(forced ? TRX2_NS_CAP : ns_lim, n));
      ns[(size_t)a] = (unsigned char)n; tot += n;
    }
    return tot;
  };
  long tot = make(target);
  if (!forced) {
    for (int it = 0; it < 16 && tot * groups < 450 && target > 2.0 * pw; it++) { target *= 0.85; tot = make(target); }
    for (int it = 0; it < 16 && tot * groups > 2000 && tot > L; it++) { target *= 1.2; tot = make(target); }
  }
  std::vector<uint2> items;
  items.reserve((size_t)tot);
  for (int a = 0; a < L; a++) {
    const int n = ns[(size_t)a], cnt = ctx->h_row_cnt[(size_t)a];
    for (int q = 0; q < n; q++) {  // the slice's bounds in the row's list travel with the item (kernel_pair.h)
      const unsigned e_lo = (unsigned)((cnt * q) / n), e_hi = (unsigned)((cnt * (q + 1)) / n);
      items.push_back(make_uint2((unsigned)a | ((unsigned)q << 10) | ((unsigned)n << 14), e_lo | (e_hi << 16)));
    }
  }
  std::stable_sort(items.begin(), items.end(), [&](const uint2& x, const uint2& y) {   // longest slices first (the launch ends with its last workgroup)
    return (x.y >> 16) - (x.y & 0xffffu) > (y.y >> 16) - (y.y & 0xffffu);
  });
  HIPCHK(hipStreamSynchronize(ctx->stream));  // nothing in flight reads the old plan while it is replaced
  if (items.size() > P.cap_items) {
    if (P.items) (void)hipFree(P.items);
    P.items = nullptr; P.cap_items = 0;
    HIPCHK(hipMalloc((void**)&P.items, sizeof(uint2) * (size_t)L * TRX2_NS_CAP));
    P.cap_items = (size_t)L * TRX2_NS_CAP;
  }
  if ((size_t)L > P.cap_L) {
    if (P.nslice) (void)hipFree(P.nslice);
    P.nslice = nullptr; P.cap_L = 0;
    HIPCHK(hipMalloc((void**)&P.nslice, (size_t)L));
    P.cap_L = (size_t)L;
  }
  HIPCHK(hipMemcpy(P.items, items.data(), sizeof(uint2) * items.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(P.nslice, ns.data(), (size_t)L, hipMemcpyHostToDevice));
  P.n_items = (int)items.size(); P.ns_max = 1;
  for (int a = 0; a < L; a++) P.ns_max = std::max(P.ns_max, (int)ns[(size_t)a]);
  P.ns_avg = (double)tot / L; P.epoch = ctx->rows_epoch; P.forced = forced;
  ctx->alloc_epoch++;  // a captured graph holds the old pointers / grid
  *index = k;
  return 0;
}

// the relax stage's re-selections from the generated restraints and their probabilities (after k_build_tables*)
static void launch_relax_masks(trx2_ctx* ctx) {
  RelaxSelArgs R;
  R.L = ctx->L; R.gen = ctx->gen; R.pd = ctx->pd; R.po = ctx->po; R.pt = ctx->pt; R.pp = ctx->pp; R.gly = ctx->gly;
  R.pc1 = TRX2_RELAX_PCUT1; R.pc2 = TRX2_RELAX_PCUT2; R.mask_r1 = ctx->mask_r1; R.mask_r2 = ctx->mask_r2;
  const size_t LL = (size_t)ctx->L * ctx->L;
  hipLaunchKernelGGL(k_relax_masks, dim3((unsigned)((LL + 255) / 256)), dim3(256), 0, ctx->stream, R);
}

// restraint tables, selection masks and packed masks from the resident distograms (ctx->cur) and parameters (ctx->prm)
static int build_tables(trx2_ctx* ctx) {
  const int L = ctx->L;
  const size_t LL = (size_t)L * L;
  const trx2_params* prm = &ctx->prm;
  if (ctx->rst_kind == 2) {  // gen_rst_af2: the 64-bin map in cur[0]
    HIPCHK(hipMemsetAsync(ctx->Td, 0, LL * TRX2_KD_AF2 * sizeof(float2), ctx->stream));
    BuildAf2Args A;
    A.L = L; A.dist = ctx->cur[0];
    A.ebase = prm->ebase; for (int k = 0; k < 3; k++) A.erep[k] = prm->erep[k];
    A.meff = prm->meff; A.pcut = prm->pcut;
    A.bk_last = std::pow(ctx->knots_af2_last / prm->dcut, prm->alpha);
    A.knots = ctx->knots_d; A.Td = ctx->Td; A.pd = ctx->pd; A.gen = ctx->gen; A.sel = ctx->sel;
    hipLaunchKernelGGL(k_build_tables_af2, dim3((unsigned)((LL + 127) / 128)), dim3(128), 0, ctx->stream, A);
    hipLaunchKernelGGL(k_pack_masks, dim3((unsigned)((LL + 255) / 256)), dim3(256), 0, ctx->stream, L, ctx->sel, (const unsigned char*)nullptr,
                       ctx->mask2, (unsigned char*)nullptr);
    launch_relax_masks(ctx);
    hipLaunchKernelGGL(k_build_rows, dim3((unsigned)L), dim3(256), 0, ctx->stream, L, (const unsigned char*)ctx->mask2, (const unsigned char*)nullptr,
                       (const unsigned char*)ctx->mask_r1, (const unsigned char*)ctx->mask_r2, ctx->rows, ctx->rows_rx, ctx->row_cnt);
    HIPCHK(hipGetLastError());
    return fetch_row_counts(ctx);
  }
  HIPCHK(hipMemsetAsync(ctx->Td, 0, LL * KD * sizeof(float2), ctx->stream));
  if (ctx->use_orient) {
    HIPCHK(hipMemsetAsync(ctx->To, 0, LL * KO * sizeof(float2), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->Tt, 0, LL * KO * sizeof(float2), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->Tp, 0, LL * KP * sizeof(float2), ctx->stream));
  }
  BuildArgs A;
  A.L = L; A.use_orient = ctx->use_orient;
  A.dist = ctx->cur[0]; A.omega = ctx->cur[1]; A.theta = ctx->cur[2]; A.phi = ctx->cur[3];
  A.ebase = prm->ebase; for (int k = 0; k < 3; k++) A.erep[k] = prm->erep[k];
  A.meff = prm->meff; A.pcut = prm->pcut;
  for (int k = 0; k < 32; k++) A.bkgr[k] = std::pow((4.25 + prm->dstep * k) / prm->dcut, prm->alpha);
  A.knots = ctx->knots_d;
  A.Td = ctx->Td; A.To = ctx->To; A.Tt = ctx->Tt; A.Tp = ctx->Tp;
  A.pd = ctx->pd; A.po = ctx->po; A.pt = ctx->pt; A.pp = ctx->pp; A.gen = ctx->gen; A.sel = ctx->sel;
  A.idr = ctx->idr; A.kind = ctx->rst_kind; A.idr_bk = ctx->idr_bk;
  hipLaunchKernelGGL(k_build_tables, dim3((unsigned)((LL + 127) / 128)), dim3(128), 0, ctx->stream, A);
  hipLaunchKernelGGL(k_pack_masks, dim3((unsigned)((LL + 255) / 256)), dim3(256), 0, ctx->stream, L, ctx->sel, (const unsigned char*)ctx->idr,
                     ctx->mask2, ctx->mask_odr);
  launch_relax_masks(ctx);
  hipLaunchKernelGGL(k_build_rows, dim3((unsigned)L), dim3(256), 0, ctx->stream, L, (const unsigned char*)ctx->mask2, (const unsigned char*)ctx->mask_odr,
                     (const unsigned char*)ctx->mask_r1, (const unsigned char*)ctx->mask_r2, ctx->rows, ctx->rows_rx, ctx->row_cnt);
  HIPCHK(hipGetLastError());
  return fetch_row_counts(ctx);
}

static int set_map_impl(trx2_ctx* ctx, int L, const char* seq, const float* dist, const float* omega, const float* theta,
                        const float* phi, const trx2_params* prm, bool device_ptrs, const unsigned char* idr = nullptr, int rst_kind = 0,
                        const double* af2_edges = nullptr) {
  if (!ctx) return 1;
  if (L < 4 || L > 1024 || !dist || !prm) { ctx->err = "trx2_set_map: need 4 <= L <= 1024, dist and params"; return 1; }
  const bool orient = omega && theta && phi;
  if (!orient && (omega || theta || phi)) { ctx->err = "trx2_set_map: omega/theta/phi must be all given or all NULL"; return 1; }
  if (rst_kind < 0 || rst_kind > 2 || (rst_kind == 1 && !idr) || (rst_kind == 2 && (orient || !af2_edges || idr))) {
    ctx->err = "trx2_set_map: restraint kind 1 (idp) needs the idr mask; kind 2 (af2) takes a 64-bin distance map with its 63 edges and no angles";
    return 1;
  }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  free_map(ctx);
  ctx->alloc_epoch++;
  const size_t LL = (size_t)L * L;
  ctx->L = L; ctx->use_orient = orient; ctx->seq = seq ? std::string(seq, strnlen(seq, L)) : std::string();
  ctx->rst_kind = rst_kind; ctx->kd = rst_kind == 2 ? TRX2_KD_AF2 : KD; ctx->dist_ca = rst_kind == 2;
  const int KDr = ctx->kd, KT = KDr + 2 * KO + KP;
  // knot positions after the reference's "%.3f" / "%.5f" text round trip (utils_ros.py:70,92,111,135)
  double* kn = ctx->knots_h;
  if (rst_kind == 2) {  // utils_ros.py:174-181: DREP = 0 / 2.325 / 3.575, then the bin edges 5 .. 61
    const double drep_af2[3] = {0.0, 2.325, 3.575};
    for (int k = 0; k < 3; k++) kn[k] = round_txt(drep_af2[k], 3);
    for (int k = 0; k < TRX2_KD_AF2 - 3; k++) kn[3 + k] = round_txt(af2_edges[5 + k], 3);
    ctx->knots_af2_last = af2_edges[5 + TRX2_KD_AF2 - 4];
  } else {
    for (int k = 0; k < 3; k++) kn[k] = round_txt(prm->drep[k], 3);
    for (int k = 0; k < 32; k++) kn[3 + k] = round_txt(4.25 + prm->dstep * k, 3);
  }
  const double astep = prm->astep_deg * M_PI / 180.0;
  {
    double start = -M_PI - 1.5 * astep, stop = M_PI + 1.5 * astep, step = (stop - start) / (KO - 1);
    for (int k = 0; k < KO; k++) {
      double v = (k == KO - 1) ? stop : start + k * step;
      kn[KDr + k] = round_txt(v, 5);
      kn[KDr + KO + k] = round_txt(v, 3);
    }
    start = -1.5 * astep; stop = M_PI + 1.5 * astep; step = (stop - start) / (KP - 1);
    for (int k = 0; k < KP; k++) kn[KDr + 2 * KO + k] = round_txt((k == KP - 1) ? stop : start + k * step, 3);
  }
  float knf[TRX2_KTOT_MAX];
  for (int k = 0; k < KT; k++) knf[k] = (float)kn[k];
  HIPCHK(hipMalloc((void**)&ctx->knots_d, sizeof(double) * TRX2_KTOT_MAX));
  HIPCHK(hipMalloc((void**)&ctx->knots_f, sizeof(float) * TRX2_KTOT_MAX));
  HIPCHK(hipMemcpyAsync(ctx->knots_d, kn, sizeof(double) * KT, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->knots_f, knf, sizeof(float) * KT, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMalloc((void**)&ctx->Td, LL * KDr * sizeof(float2)));
  if (idr) {  // pair flags: table variant (kind 1) and / or the ordered-pairs stage of mode 3
    HIPCHK(hipMalloc((void**)&ctx->idr, LL));
    HIPCHK(hipMalloc((void**)&ctx->mask_odr, LL));
    HIPCHK(hipMemcpyAsync(ctx->idr, idr, LL, hipMemcpyHostToDevice, ctx->stream));
    std::vector<double> bk(32 * 32);
    for (int m = 0; m < 32; m++)
      for (int k = 0; k < 32; k++) bk[m * 32 + k] = std::pow((4.25 + prm->dstep * k) / (4.25 + prm->dstep * m), prm->alpha);
    HIPCHK(hipMalloc((void**)&ctx->idr_bk, sizeof(double) * 32 * 32));
    HIPCHK(hipMemcpy(ctx->idr_bk, bk.data(), sizeof(double) * 32 * 32, hipMemcpyHostToDevice));
  }
  HIPCHK(hipMalloc((void**)&ctx->pd, LL * 4));
  HIPCHK(hipMalloc((void**)&ctx->gen, LL));
  HIPCHK(hipMalloc((void**)&ctx->sel, LL));
  HIPCHK(hipMalloc((void**)&ctx->mask2, LL));
  HIPCHK(hipMalloc((void**)&ctx->rows, LL * sizeof(unsigned)));
  HIPCHK(hipMalloc((void**)&ctx->row_cnt, (size_t)L * sizeof(int)));
  HIPCHK(hipMalloc((void**)&ctx->rows_rx, LL * sizeof(unsigned short)));
  HIPCHK(hipMalloc((void**)&ctx->mask_r1, LL));
  HIPCHK(hipMalloc((void**)&ctx->mask_r2, LL));
  {  // glycines: the relax stage's re-selections drop every pair that touches one (utils_ros.py:713-717)
    std::vector<unsigned char> gg((size_t)L);
    for (int i = 0; i < L; i++) gg[(size_t)i] = (i < (int)ctx->seq.size() && ctx->seq[(size_t)i] == 'G') ? 1 : 0;
    HIPCHK(hipMalloc((void**)&ctx->gly, (size_t)L));
    HIPCHK(hipMemcpy(ctx->gly, gg.data(), (size_t)L, hipMemcpyHostToDevice));
  }
  {  // residues that donate a backbone hydrogen bond: every residue with a predecessor except proline (trx2_model.h)
    std::vector<unsigned char> hh((size_t)L);
    for (int i = 0; i < L; i++) hh[i] = (i >= 1 && !(i < (int)ctx->seq.size() && ctx->seq[i] == 'P')) ? 1 : 0;
    // ... and behind the flags, 256-byte aligned, the per-residue parameter block of the rama term (trx2_model.h trx2_rama_params;
    // kernel_step.h rama_par_ptr): one allocation, so that every owner / borrower of hasH owns / borrows both
    const size_t off = ((size_t)L + 255) & ~(size_t)255;
    std::vector<unsigned char> blk(off + (size_t)L * TRX2_RAMA_NPAR * sizeof(float), 0);
    memcpy(blk.data(), hh.data(), (size_t)L);
    const std::string sq = ctx->seq.size() >= (size_t)L ? ctx->seq : std::string((size_t)L, 'A');
    for (int i = 0; i < L; i++) trx2_rama_params(sq.c_str(), i, L, reinterpret_cast<float*>(blk.data() + off) + (size_t)i * TRX2_RAMA_NPAR);
    if (const char* e = getenv("TRX2_RAMA_SCAN")) {   // model scans only (tools/tol_sweep_relax.py): "surface scale,helix scale" on the fitted parts
      float sf = 1.0f, sh = 1.0f;
      if (sscanf(e, "%f,%f", &sf, &sh) == 2)
        for (int i = 0; i < L; i++) {
          float* p = reinterpret_cast<float*>(blk.data() + off) + (size_t)i * TRX2_RAMA_NPAR;
          for (int k = 1; k <= 8; k++) p[k] *= sf;
          p[9] *= sh;
        }
    }
    HIPCHK(hipMalloc((void**)&ctx->hasH, blk.size()));
    HIPCHK(hipMemcpy(ctx->hasH, blk.data(), blk.size(), hipMemcpyHostToDevice));
  }
  if (orient) {
    HIPCHK(hipMalloc((void**)&ctx->To, LL * KO * sizeof(float2)));
    HIPCHK(hipMalloc((void**)&ctx->Tt, LL * KO * sizeof(float2)));
    HIPCHK(hipMalloc((void**)&ctx->Tp, LL * KP * sizeof(float2)));
    HIPCHK(hipMalloc((void**)&ctx->po, LL * 4));
    HIPCHK(hipMalloc((void**)&ctx->pt, LL * 4));
    HIPCHK(hipMalloc((void**)&ctx->pp, LL * 4));
  }
  const float* src[4] = {dist, omega, theta, phi};
  const int nb[4] = {rst_kind == 2 ? 64 : TRX2_ND_BINS, TRX2_NO_BINS, TRX2_NO_BINS, TRX2_NP_BINS};
  for (int c = 0; c < 4; c++) {  // the distograms stay resident: the feedback step re-weights them in place
    if (!src[c]) continue;
    HIPCHK(hipMalloc((void**)&ctx->cur[c], LL * nb[c] * 4));
    HIPCHK(hipMemcpyAsync(ctx->cur[c], src[c], LL * nb[c] * 4, device_ptrs ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
  }
  ctx->prm = *prm;
  if (build_tables(ctx)) return 1;
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
extern "C" int trx2_set_map_ex(trx2_ctx* ctx, int L, const char* seq, const float* dist, const float* omega, const float* theta,
                               const float* phi, const trx2_params* prm, const unsigned char* idr, int rst_kind) {
  if (rst_kind == 2) { if (ctx) ctx->err = "trx2_set_map_ex: use trx2_set_map_af2 for AlphaFold-style maps"; return 1; }
  return set_map_impl(ctx, L, seq, dist, omega, theta, phi, prm, false, idr, rst_kind);
}
extern "C" int trx2_set_map_af2(trx2_ctx* ctx, int L, const char* seq, const float* dist64, const double* edges63, const trx2_params* prm) {
  return set_map_impl(ctx, L, seq, dist64, nullptr, nullptr, nullptr, prm, false, nullptr, 2, edges63);
}
// replace the values of n table rows (gen_gpcr_rst's edits of the flagged pairs, utils_ros.py:551,579,601,627)
extern "C" int trx2_override_table_rows(trx2_ctx* ctx, int channel, int n, const int* a, const int* b, const double* y) {
  if (!ctx || !ctx->L) return 1;
  if (channel < 0 || channel > 3 || (channel > 0 && !ctx->use_orient) || n < 0 || (n && (!a || !b || !y))) { ctx->err = "trx2_override_table_rows: bad arguments"; return 1; }
  if (!n) return 0;
  const int L = ctx->L;
  for (int i = 0; i < n; i++)
    if (a[i] < 0 || a[i] >= L || b[i] < 0 || b[i] >= L || a[i] == b[i] || (channel < 2 && a[i] > b[i])) {
      ctx->err = "trx2_override_table_rows: pair out of range (dist / omega rows live at a < b)"; return 1;
    }
  HIPCHK(hipSetDevice(ctx->device));
  const int K = channel == 0 ? ctx->kd : channel == 3 ? KP : KO;
  const int off = channel == 0 ? 0 : channel == 1 ? ctx->kd : channel == 2 ? ctx->kd + KO : ctx->kd + 2 * KO;
  float2* T = channel == 0 ? ctx->Td : channel == 1 ? ctx->To : channel == 2 ? ctx->Tt : ctx->Tp;
  int *da = nullptr, *db = nullptr; double* dy = nullptr;
  HIPCHK(hipMalloc((void**)&da, sizeof(int) * n)); HIPCHK(hipMalloc((void**)&db, sizeof(int) * n)); HIPCHK(hipMalloc((void**)&dy, sizeof(double) * (size_t)n * K));
  HIPCHK(hipMemcpy(da, a, sizeof(int) * n, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(db, b, sizeof(int) * n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dy, y, sizeof(double) * (size_t)n * K, hipMemcpyHostToDevice));
  if (ctx->child && ctx->child->stream) HIPCHK(hipStreamSynchronize(ctx->child->stream));
  hipLaunchKernelGGL(k_override_rows, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, n, K, L, da, db, dy, ctx->knots_d + off, T);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dy);
  ctx->tab_epoch++;   // cached segments of the edited rows are stale
  if (ctx->child) ctx->child->tab_epoch++;
  return 0;
}

extern "C" int trx2_get_tables(trx2_ctx* ctx, int channel, float* y_y2, float* knots, float* prob, unsigned char* gen,
                               unsigned char* sel) {
  if (!ctx || !ctx->L) return 1;
  if (channel < 0 || channel > 3 || (channel > 0 && !ctx->use_orient)) { ctx->err = "trx2_get_tables: bad channel"; return 1; }
  HIPCHK(hipSetDevice(ctx->device));
  const size_t LL = (size_t)ctx->L * ctx->L;
  const int K[4] = {ctx->kd, KO, KO, KP};
  const int off[4] = {0, ctx->kd, ctx->kd + KO, ctx->kd + 2 * KO};
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
  // row plans of every launch shape this batch can take: its own, and those the tail compaction shrinks through (one decoy group
  // of 64 fewer at a time, then 32, 16, .. 1 decoys per wave); the record buffer must hold the widest of them
  int plan_idx = -1;
  if (plan_get(ctx, 64 / BW, ngrp, &plan_idx)) return 1;
  size_t fa_units = (size_t)ctx->plans[(size_t)plan_idx].ns_max * B;
  for (int g = 1; g < ngrp; g++) {
    int k;
    if (plan_get(ctx, 1, g, &k)) return 1;
    fa_units = std::max(fa_units, (size_t)ctx->plans[(size_t)k].ns_max * g * 64);
  }
  for (int w = 1; w < 64; w <<= 1)
    if (w < BW || ngrp > 1) {
      int k;
      if (plan_get(ctx, 64 / w, 1, &k)) return 1;
      fa_units = std::max(fa_units, (size_t)ctx->plans[(size_t)k].ns_max * w);
    }
  const bool layout_changed = BW != ctx->BW || Bpad != ctx->Bpad;
  ctx->BW = BW; ctx->Bpad = Bpad; ctx->plan_cur = plan_idx;
  if (B <= ctx->Bcap && L <= ctx->Lcap && fa_units * L <= ctx->fa_cap) {
    if (layout_changed) {  // the pad lanes of a group layout must hold finite numbers
      HIPCHK(hipMemsetAsync(ctx->xyzT, 0, sizeof(float4) * (size_t)Bpad * L * 5, ctx->stream));
      ctx->alloc_epoch++;
    }
    return 0;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  free_batch(ctx);
  const size_t BL = (size_t)B * L;
  HIPCHK(hipMalloc((void**)&ctx->st_i, sizeof(int) * B * SI_N));
  HIPCHK(hipMalloc((void**)&ctx->st_d, sizeof(double) * B * SD_N));
  HIPCHK(hipMalloc((void**)&ctx->rho, sizeof(float) * B * LBM));
  HIPCHK(hipMalloc((void**)&ctx->gram, sizeof(double) * B * GR_N));
  HIPCHK(hipMemsetAsync(ctx->gram, 0, sizeof(double) * B * GR_N, ctx->stream));
  HIPCHK(hipMalloc((void**)&ctx->X, sizeof(float4) * BL));
  HIPCHK(hipMalloc((void**)&ctx->G, sizeof(float4) * BL));
  HIPCHK(hipMalloc((void**)&ctx->D, sizeof(float4) * BL));
  HIPCHK(hipMalloc((void**)&ctx->XT, sizeof(float4) * BL));
  HIPCHK(hipMalloc((void**)&ctx->S, sizeof(float4) * BL * LBM));
  HIPCHK(hipMalloc((void**)&ctx->Y, sizeof(float4) * BL * LBM));
  HIPCHK(hipMalloc((void**)&ctx->P, sizeof(float4) * BL * 5));
  HIPCHK(hipMalloc((void**)&ctx->geom, sizeof(float4) * BL * 3));
  // the decoy-minor copy is sized for the widest group layout of this many decoys, so that any later, smaller batch fits
  const size_t xt_alloc = (size_t)((B + 63) / 64 * 64) * L * 5;
  HIPCHK(hipMalloc((void**)&ctx->xyzT, sizeof(float4) * xt_alloc));
  HIPCHK(hipMalloc((void**)&ctx->wcur, sizeof(float) * B * 8));
  HIPCHK(hipMalloc((void**)&ctx->FA, sizeof(float) * fa_units * L * PR_REC));
  HIPCHK(hipMalloc((void**)&ctx->plan, sizeof(int) * (1 + 2 * 64)));
  HIPCHK(hipMalloc((void**)&ctx->e_last, sizeof(double) * B * TRX2_NTERMS));
  HIPCHK(hipMalloc((void**)&ctx->f_last, sizeof(double) * B));
  HIPCHK(hipMalloc((void**)&ctx->grad, sizeof(float) * BL * 3));
  HIPCHK(hipMalloc((void**)&ctx->tors0, sizeof(float) * BL * 3));
  HIPCHK(hipMalloc((void**)&ctx->done_count, sizeof(int)));
  HIPCHK(hipMalloc((void**)&ctx->seq_ctr, sizeof(int)));
  HIPCHK(hipMemsetAsync(ctx->seq_ctr, 0, sizeof(int), ctx->stream));
  HIPCHK(hipMalloc((void**)&ctx->runs, sizeof(trx2_run) * TRX2_MAX_RUNS));
  HIPCHK(hipMemsetAsync(ctx->P, 0, sizeof(float4) * BL * 5, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->xyzT, 0, sizeof(float4) * xt_alloc, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->wcur, 0, sizeof(float) * B * 8, ctx->stream));
  ctx->Bcap = B; ctx->Lcap = L; ctx->fa_cap = fa_units * L;
  ctx->alloc_epoch++;
  return 0;
}

// Segment cache of the pair kernel (kernel_pair.h, PairArgs / SEGC) for SINGLE-decoy folds: one block of NSEG cubic segments + their
// tags per row entry, rows at a fixed stride of L entries (2.3 MB at 150 residues with all channels, 16.6 MB at 400).  (Re)allocated
// when the map outgrows it; the tags are reset whenever the tables have changed since (a new map, a feedback step, edited rows).
// TRX2_SEG_CACHE=0 turns it off (A/B timing; results are bit-identical either way).
static int ensure_seg_cache(trx2_ctx* ctx) {
  static const bool off = getenv("TRX2_SEG_CACHE") && atoi(getenv("TRX2_SEG_CACHE")) == 0;
  if (off || ctx->Bpad != 1) return 0;   // batches keep their gathers (the lanes of a wave share table rows there); seg_on() says which
  const size_t L = (size_t)ctx->L, nseg = ctx->use_orient ? 6 : 1;
  const size_t elems = L * L, bytes = elems * nseg * sizeof(float4);
  if (bytes > ctx->seg_bytes || elems > ctx->seg_elems) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->segc) (void)hipFree(ctx->segc);
    if (ctx->segt) (void)hipFree(ctx->segt);
    ctx->segc = nullptr; ctx->segt = nullptr; ctx->seg_bytes = ctx->seg_elems = 0;
    HIPCHK(hipMalloc((void**)&ctx->segc, bytes));
    HIPCHK(hipMalloc((void**)&ctx->segt, elems * sizeof(uint2)));
    ctx->seg_bytes = bytes; ctx->seg_elems = elems; ctx->seg_epoch = -1;
    ctx->alloc_epoch++;
  }
  if (ctx->seg_epoch != ctx->tab_epoch) {   // 0xff in every tag byte: no segment has that index
    HIPCHK(hipMemsetAsync(ctx->segt, 0xff, elems * sizeof(uint2), ctx->stream));
    ctx->seg_epoch = ctx->tab_epoch;
  }
  return 0;
}
// the launch about to be made uses the cache: a single-decoy shape whose blocks and tags are in place for the current tables
static bool seg_on(const trx2_ctx* c) { return c->BW == 1 && c->Bpad == 1 && c->segc && c->seg_epoch == c->tab_epoch && c->seg_elems >= (size_t)c->L * c->L; }

static PairArgs pair_args(trx2_ctx* c, int B) {
  PairArgs P;
  P.L = c->L; P.B = B; P.Bpad = c->Bpad; P.items = c->plans[(size_t)c->plan_cur].items; P.n_items = c->plans[(size_t)c->plan_cur].n_items;
  P.xyzT = c->xyzT; P.Td = c->Td; P.To = c->To; P.Tt = c->Tt; P.Tp = c->Tp;
  P.rows = c->rows; P.row_cnt = c->row_cnt; P.rows_rx = c->rows_rx; P.has_odr = c->mask_odr != nullptr;
  P.knots = c->knots_f; P.wcur = c->wcur; P.FA = c->FA; P.seq_ctr = c->seq_ctr;
  P.kd = c->kd; P.dist_ca = c->dist_ca;
  const bool sc = seg_on(c);
  P.segc = sc ? c->segc : nullptr; P.segt = sc ? c->segt : nullptr; P.rowcap = c->L;
  return P;
}
static ChainArgs chain_args(trx2_ctx* c, int B, int mode, int nruns, int max_evals) {
  ChainArgs A;
  A.seq_ctr = c->seq_ctr;
  A.L = c->L; A.B = B; A.mode = mode; A.nruns = nruns;
  A.max_evals = max_evals; A.runs = c->runs; A.st_i = c->st_i; A.st_d = c->st_d; A.rho = c->rho; A.gram = c->gram;
  A.X = c->X; A.G = c->G; A.D = c->D; A.XT = c->XT; A.S = c->S; A.Y = c->Y; A.P = c->P; A.geom = c->geom;
  A.xyzT = c->xyzT; A.BW = c->BW;
  A.wcur = c->wcur; A.FA = c->FA; A.nslice = c->plans[(size_t)c->plan_cur].nslice; A.hasH = c->hasH; A.ns_max = c->plans[(size_t)c->plan_cur].ns_max;
  A.e_last = c->e_last; A.f_last = c->f_last;
  A.grad_out = c->grad; A.done_count = c->done_count;
  A.slot_id = c->slot_id; A.next_id = c->next_id; A.n_total = 0; A.seed = 0; A.decoy0 = 0; A.tors0_all = nullptr;
  A.out_xyz = c->out_xyz; A.out_X = c->out_X; A.out_e = c->out_e; A.out_f = c->out_f; A.out_stat = c->out_stat;
  return A;
}
// Returns non-zero (with c->err set) when the launch was REFUSED: the caller must stop enqueuing -- a step kernel launched after a
// refused pair launch would sum stale records of an earlier evaluation and minimise on them.
static int launch_pair(trx2_ctx* c, int B) {
  // never launch a shape the buffers were not sized for (operand shapes are checked on the host: a kernel that writes out of
  // bounds can reset every GPU of the node)
  const RowPlan* rp = (c->plan_cur >= 0 && c->plan_cur < (int)c->plans.size()) ? &c->plans[(size_t)c->plan_cur] : nullptr;
  if (!rp || rp->epoch != c->rows_epoch || rp->pw != 64 / c->BW || rp->groups < c->Bpad / c->BW || rp->n_items < c->L || B < 1 || B > c->Bpad ||
      B > c->Bcap || c->Bpad % c->BW != 0 || (size_t)rp->ns_max * B * c->L > c->fa_cap || (size_t)c->Bpad > (size_t)(c->Bcap + 63) / 64 * 64) {
    c->err = "internal: pair-kernel launch shape does not fit the batch buffers or its row plan";
    fprintf(stderr, "trx2fold: %s (B=%d Bpad=%d Bcap=%d BW=%d plan=%d)\n", c->err.c_str(), B, c->Bpad, c->Bcap, c->BW, c->plan_cur);
    return 1;
  }
  const PairArgs P = pair_args(c, B);
  const dim3 grid((unsigned)rp->n_items, 1, c->Bpad / c->BW), block(PAIR_THREADS);
  // maps without the angle channels (--no-orient, gen_rst_af2) run the instantiation without the angular block
#define LAUNCH_PAIR(W)                                                                                       \
  if (c->use_orient) hipLaunchKernelGGL((k_pair<W, FAM_ALL>), grid, block, 0, c->stream, P);               \
  else hipLaunchKernelGGL((k_pair<W, FAM_DIST | FAM_VDW>), grid, block, 0, c->stream, P)
  // a single decoy: one wave per row on a context set so (trx2_ctx_set_single_decoy_waves), with the segment cache where it is in place
  if (c->BW == 1 && (c->pair1_waves == 1 || P.segc)) {
    const bool ang = c->use_orient != 0;
    if (c->pair1_waves == 1) {
      if (P.segc) { if (ang) hipLaunchKernelGGL((k_pair1<FAM_ALL, true>), grid, dim3(64), 0, c->stream, P); else hipLaunchKernelGGL((k_pair1<FAM_DIST | FAM_VDW, true>), grid, dim3(64), 0, c->stream, P); }
      else { if (ang) hipLaunchKernelGGL((k_pair1<FAM_ALL, false>), grid, dim3(64), 0, c->stream, P); else hipLaunchKernelGGL((k_pair1<FAM_DIST | FAM_VDW, false>), grid, dim3(64), 0, c->stream, P); }
    } else {
      if (ang) hipLaunchKernelGGL((k_pair_c<FAM_ALL>), grid, block, 0, c->stream, P); else hipLaunchKernelGGL((k_pair_c<FAM_DIST | FAM_VDW>), grid, block, 0, c->stream, P);
    }
    return 0;
  }
  switch (c->BW) {
    case 64: LAUNCH_PAIR(64); break;
    case 32: LAUNCH_PAIR(32); break;
    case 16: LAUNCH_PAIR(16); break;
    case 8: LAUNCH_PAIR(8); break;
    case 4: LAUNCH_PAIR(4); break;
    case 2: LAUNCH_PAIR(2); break;
    default: LAUNCH_PAIR(1); break;
  }
#undef LAUNCH_PAIR
  return 0;
}
static CartArgs cart_args(trx2_ctx* c, int B, int nruns, int max_evals) {
  CartArgs A;
  A.L = c->L; A.B = B; A.nruns = nruns; A.max_evals = max_evals; A.seq_ctr = c->seq_ctr;
  A.runs = c->runs; A.st_i = c->st_i; A.st_d = c->st_d; A.rho = c->rho; A.gram = c->gram;
  A.CX = c->CX; A.CG = c->CG; A.CD = c->CD; A.CS = c->CS; A.CY = c->CY;
  A.P = c->P; A.xyzT = c->xyzT; A.BW = c->BW; A.X = c->X; A.XT = c->XT; A.geom = c->geom; A.wcur = c->wcur;
  A.FA = c->FA; A.nslice = c->plans[(size_t)c->plan_cur].nslice; A.hasH = c->hasH; A.ns_max = c->plans[(size_t)c->plan_cur].ns_max;
  A.e_last = c->e_last; A.f_last = c->f_last; A.done_count = c->done_count;
  A.hist_lds = 0;
  return A;
}
static void launch_chain_args(trx2_ctx* c, int B, const ChainArgs& A);
static void launch_chain(trx2_ctx* c, int B, int mode, int nruns, int max_evals) {
  launch_chain_args(c, B, chain_args(c, B, mode, nruns, max_evals));
}
static void launch_chain_args(trx2_ctx* c, int B, const ChainArgs& A) {
  const dim3 grid(B);
  const int L = c->L;
  // chains of up to 128 residues (the reference's example has 90) run the step on two waves: every workgroup reduction and
  // scan combines two partials instead of four
  if (L <= 128) hipLaunchKernelGGL((k_chain<1, 128>), grid, dim3(128), HIST_LDS_BYTES(L), c->stream, A);
  else if (L <= CHAIN_THREADS) hipLaunchKernelGGL((k_chain<1, CHAIN_THREADS>), grid, dim3(CHAIN_THREADS), HIST_LDS_BYTES(L), c->stream, A);
  else if (L <= 2 * CHAIN_THREADS) hipLaunchKernelGGL((k_chain<1, 2 * CHAIN_THREADS>), grid, dim3(2 * CHAIN_THREADS), HIST_LDS_BYTES(L), c->stream, A);  // one residue per thread
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
  if (ensure_batch(ctx, B) || ensure_seg_cache(ctx)) return 1;
  const int L = ctx->L;
  const size_t BL = (size_t)B * L;
  if (upload_single_run(ctx, w, sep_lo, sep_hi)) return 1;
  HIPCHK(hipMemcpyAsync(ctx->tors0, tors, sizeof(float) * BL * 3, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->st_i, 0, sizeof(int) * B * SI_N, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->st_d, 0, sizeof(double) * B * SD_N, ctx->stream));
  hipLaunchKernelGGL(k_init_torsions, dim3((unsigned)((BL + 255) / 256)), dim3(256), 0, ctx->stream, L, B, 0ull, 0u,
                     ctx->tors0, ctx->X, ctx->XT, ctx->geom);
  launch_chain(ctx, B, MODE_INIT, 1, 1 << 30);
  if (launch_pair(ctx, B)) { (void)hipStreamSynchronize(ctx->stream); return 1; }
  launch_chain(ctx, B, MODE_FINISH, 1, 1 << 30);
  HIPCHK(hipGetLastError());
  if (e_terms) HIPCHK(hipMemcpyAsync(e_terms, ctx->e_last, sizeof(double) * B * TRX2_NTERMS, hipMemcpyDeviceToHost, ctx->stream));
  if (f_total) HIPCHK(hipMemcpyAsync(f_total, ctx->f_last, sizeof(double) * B, hipMemcpyDeviceToHost, ctx->stream));
  if (grad) HIPCHK(hipMemcpyAsync(grad, ctx->grad, sizeof(float) * BL * 3, hipMemcpyDeviceToHost, ctx->stream));
  std::vector<float> tmp;
  if (xyz) {
    tmp.resize(BL * 20);
    HIPCHK(hipMemcpyAsync(tmp.data(), ctx->P, sizeof(float) * BL * 20, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (xyz)
    for (size_t i = 0; i < BL; i++) memcpy(xyz + i * 15, tmp.data() + i * 20, 15 * sizeof(float));
  return 0;
}

static int ensure_outputs(trx2_ctx* ctx, size_t N, bool with_tors0) {
  const size_t L = (size_t)(ctx->L > ctx->Lcap ? ctx->L : ctx->Lcap);  // sized like the batch buffers: a later, longer map that fits them fits these
  if (N > ctx->out_cap || L > ctx->out_L || !ctx->slot_id) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    void* q[] = {ctx->slot_id, ctx->next_id, ctx->out_stat, ctx->out_xyz, ctx->out_X, ctx->out_e, ctx->out_f, ctx->tors0_all};
    for (void* v : q)
      if (v) (void)hipFree(v);
    ctx->tors0_all = nullptr;
    const size_t cap = N > (size_t)ctx->Bcap ? N : (size_t)ctx->Bcap;
    HIPCHK(hipMalloc((void**)&ctx->slot_id, sizeof(int) * cap));
    HIPCHK(hipMalloc((void**)&ctx->next_id, sizeof(int)));
    HIPCHK(hipMalloc((void**)&ctx->out_stat, sizeof(int) * cap * 4));
    HIPCHK(hipMalloc((void**)&ctx->out_xyz, sizeof(float4) * cap * L * 4));
    HIPCHK(hipMalloc((void**)&ctx->out_X, sizeof(float4) * cap * L));
    HIPCHK(hipMalloc((void**)&ctx->out_e, sizeof(double) * cap * TRX2_NTERMS));
    HIPCHK(hipMalloc((void**)&ctx->out_f, sizeof(double) * cap));
    ctx->out_cap = cap; ctx->out_L = L;
    ctx->alloc_epoch++;
  }
  if (with_tors0 && !ctx->tors0_all) {
    HIPCHK(hipMalloc((void**)&ctx->tors0_all, sizeof(float) * ctx->out_cap * L * 3));
    ctx->alloc_epoch++;
  }
  return 0;
}

// Folds N decoys on P = min(N, pool) slots.  Every evaluation is one pair-kernel launch over the P slots and one step-kernel
// launch; a decoy that is over reports (one evaluation at its accepted point under the last run's weights), its slot then takes
// the next decoy of the queue ON THE DEVICE (kernel_step.h) -- the host only replays chunks of launches and polls the number of
// retired slots.  A decoy's identity is (seed, decoy0 + index): results do not depend on the slot that folded it.
// Dynamic LDS the fused step launch uses for the Cartesian role's staged history.  A context that is alone in the process takes
// all there is (its own pair kernel never runs beside its step kernel).  With a second lane or another chain's context the
// other stream's pair kernel does, and a step workgroup
// that fills a CU's LDS keeps it off that CU: two pair-kernel workgroups' worth (2 x 28.0 KB) stay free -- measured at 2 x 160
// slots: no reserve 837, one workgroup's 865, two 900 decoys/s (all channels 789 -> 840 from one to two); one lane of 64 slots
// loses 2 % with the reserve, hence the distinction (profiles/README.md).
static int step_dyn_budget(const trx2_ctx* ctx, int k, int L) {
  static const int env = getenv("TRX2_STEP_LDS_RESERVE") ? atoi(getenv("TRX2_STEP_LDS_RESERVE")) : -1;  // A/B timing only
  const bool shared = g_live_contexts.load() > 1;  // a second lane (a context of its own) or another chain's context in this process
  const int reserve = env >= 0 ? env : (shared ? 2 * ctx->pair_static + 1024 : 0);   // two workgroups of the other stream's pair kernel
  return std::max(HIST_LDS_BYTES(L), ctx->step_dyn_max[k] - reserve);
}
// A fold on many slots launches more step workgroups (of both lanes) than the chip has CUs, and the fused kernel's 256 + 106
// registers keep them at one per CU (640 slots: 2.5 rounds, 55 us).  Chains of 129-256 residues then fold on the low-register
// instantiation -- the same arithmetic, bit for bit (tests), one stored pair at a time in the Cartesian role, 256 registers, the
// role's arrays in dynamic LDS (9 KB static instead of 35) -- with no more dynamic LDS than lets two workgroups share a CU.  Decided once per fold, by the
// slots it starts with (the tail compaction shrinks the launches, not the choice).  TRX2_STEP_ONE_PER_CU=1 (read per fold): never
// (A/B timing, and the test that compares the two instantiations bit for bit).
static bool step_two_per_cu(const trx2_ctx* ctx, int L, int slots, int* dyn_cap) {
  // measured crossovers (tools/pool_sweep.py, 1280 decoys): chains of 129-256 residues +2 % at 160 slots per lane, +5 % at 192,
  // +11 % at 256, -3 % at 32; L=90: +1..3 % at 128, +8 % at 192, +19 % at 320, +30 % at 640, -7 % at 64 with distances only
  int min_slots = L <= 128 ? 128 : 160;
  if (const char* e = getenv("TRX2_STEP_LOWREG_MIN")) min_slots = atoi(e);   // A/B timing only
  if (getenv("TRX2_STEP_ONE_PER_CU") != nullptr || slots < min_slots || L > CHAIN_THREADS) return false;
  const int k = L <= 128 ? 0 : 1, n = L <= 128 ? 4 : 2;   // workgroups per CU: eight waves either way
  const int cap = (ctx->lds_total - n * ctx->step2_static[k]) / n / 16 * 16;
  if ((int)HIST_LDS_BYTES(L) > cap || (int)CART_ARRAYS_BYTES(L) > cap) return false;
  *dyn_cap = cap;
  return true;
}

#include "launch_engine.h"

static int fold_impl(trx2_ctx* ctx, int N, const trx2_run* runs, int nruns, uint64_t seed, uint32_t decoy0,
                     const float* tors0, int max_evals, float* tors_out, float* xyz_out, double* e_terms,
                     double* f_final, int* status, int* n_evals, int* n_iters) {
  if (!ctx) return 1;
  if (!ctx->L) { ctx->err = "trx2_fold_batch: no map set"; return 1; }
  if (N < 1 || !runs || nruns < 1 || nruns > TRX2_MAX_RUNS) { ctx->err = "trx2_fold_batch: bad arguments (need B >= 1 decoys and 1 .. " + std::to_string(TRX2_MAX_RUNS) + " runs)"; return 1; }
  bool has_cart = false, has_filter = false;
  for (int i = 0; i < nruns; i++) {
    has_cart |= runs[i].cartesian != 0; has_filter |= runs[i].pair_filter == TRX2_FILTER_ODR;
    if (runs[i].pair_filter < 0 || runs[i].pair_filter > TRX2_FILTER_RELAX2 || !(runs[i].tol >= 0.0f)) { ctx->err = "trx2_fold_batch: a run's pair_filter / tol is out of range"; return 1; }
  }
  if (has_cart && ctx->L > 2 * CHAIN_THREADS) { ctx->err = "trx2_fold_batch: Cartesian-space runs support chains of up to 512 residues"; return 1; }
  if (has_filter && !ctx->mask_odr) { ctx->err = "trx2_fold_batch: a run filters by the idr mask, but the map was set without one (trx2_set_map_ex)"; return 1; }
  if (max_evals <= 0) max_evals = 1 << 30;
  HIPCHK(hipSetDevice(ctx->device));
  const int B0 = (ctx->pool > 0 && ctx->pool < N) ? ctx->pool : N;   // slots
  int B = B0;   // shrinks by a decoy group at a time at the tail of the fold (below)
  if (ensure_batch(ctx, B) || ensure_seg_cache(ctx)) return 1;
  if (has_cart && ensure_cart(ctx, B)) return 1;
  if (ensure_outputs(ctx, (size_t)N, tors0 != nullptr)) return 1;
  // The launch shape of the full batch comes back when the fold ends, however it ends: the buffers are laid out for it, and
  // whoever launches on them next (the pair-kernel replays of trx2_time_pair_kernel, for one) sizes its grid and its record
  // indices by these fields.  (A replay of B slots on the shape the tail compaction had left behind wrote records past the end
  // of the buffer: a GPU memory fault, found under rocprofv3.)
  struct ShapeGuard { trx2_ctx* c; int bpad, plan, bw; ~ShapeGuard() { c->Bpad = bpad; c->plan_cur = plan; c->BW = bw; } } shape_guard{ctx, ctx->Bpad, ctx->plan_cur, ctx->BW};
  const int L = ctx->L;
  const size_t BL = (size_t)B * L, NL = (size_t)N * L;
  auto t0 = std::chrono::steady_clock::now();
  HIPCHK(hipMemcpyAsync(ctx->runs, runs, sizeof(trx2_run) * nruns, hipMemcpyHostToDevice, ctx->stream));
  if (tors0) HIPCHK(hipMemcpyAsync(ctx->tors0_all, tors0, sizeof(float) * NL * 3, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->st_i, 0, sizeof(int) * B * SI_N, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->st_d, 0, sizeof(double) * B * SD_N, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->gram, 0, sizeof(double) * B * GR_N, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->done_count, 0, sizeof(int), ctx->stream));
  {
    std::vector<int> ids((size_t)B);
    for (int i = 0; i < B; i++) ids[i] = i;
    HIPCHK(hipMemcpyAsync(ctx->slot_id, ids.data(), sizeof(int) * B, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->next_id, &B, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));  // ids / B leave scope
  }
  hipLaunchKernelGGL(k_init_torsions, dim3((unsigned)((BL + 255) / 256)), dim3(256), 0, ctx->stream, L, B, seed, decoy0,
                     tors0 ? ctx->tors0_all : (const float*)nullptr, ctx->X, ctx->XT, ctx->geom);
  launch_chain(ctx, B, MODE_INIT, nruns, max_evals);
  int launches = 0;
  double slot_launches = 0;  // sum over launch pairs of the slots they served
  const int chunk = 64;
  const int compact_mode = ctx->compact;
  // hard cap on launches: a decoy stops by itself at max_evals (+ its report, + skipped runs); a slot folds ceil(N / B) of them
  const long rounds = (N + B0 - 1) / B0;
  long cap = ((long)max_evals + 96) * rounds;
  if (cap > 200000000L || cap < 0) cap = 200000000L;
  HIPCHK(hipMemsetAsync(ctx->seq_ctr, 0, sizeof(int), ctx->stream));
  const int pe = ctx->prof_every;
  int prof_used = 0;
  ctx->prof_pair_ms = ctx->prof_step_ms = 0; ctx->prof_n = 0; ctx->prof_a.clear(); ctx->prof_b.clear();
  if (pe > 0 && ctx->prof_ev.empty()) {
    ctx->prof_ev.resize(3 * (size_t)chunk);
    for (auto& e : ctx->prof_ev) HIPCHK(hipEventCreate(&e));
  }
  auto pool_args = [&](ChainArgs& ca) {
    ca.n_total = N; ca.seed = seed; ca.decoy0 = decoy0; ca.tors0_all = tors0 ? ctx->tors0_all : nullptr;
  };
  int two_cap = 0;
  bool two_per_cu = has_cart && step_two_per_cu(ctx, L, B0, &two_cap);
  // Shared launches in half-evaluation form (launch_engine.h, k_half_multi): the fold's step runs in the low-register instantiation
  // inside a kernel that also carries other folds' pair work items -- its step arguments are laid out for that form from the start
  // (also when the engines turn out to be full and the fold launches for itself: k_step<.., true>, the same arithmetic).
  const bool want_engine = B == 1 && N == 1 && has_cart && L <= 2 * CHAIN_THREADS && ctx->prof_every == 0 && getenv("TRX2_GRAPH") == nullptr &&
                           engine_enabled(g_live_contexts.load() - g_lane_contexts.load());
  bool engine_half = false;
  if (want_engine && !two_per_cu && engine_half_enabled(g_live_contexts.load() - g_lane_contexts.load()) && ctx->pair1_waves == 1 && L <= CHAIN_THREADS && L <= PAIR_SUB_ENT && seg_on(ctx)) {
    const int k = L <= 128 ? 0 : 1, n = L <= 128 ? 4 : 2;
    const int cap = (ctx->lds_total - n * ctx->half_static[k]) / n / 16 * 16;
    if (ctx->half_static[k] > 0 && (int)HIST_LDS_BYTES(L) <= cap && (int)CART_ARRAYS_BYTES(L) <= cap) { two_per_cu = true; two_cap = cap; engine_half = true; }
  }
  // dynamic LDS of a step launch: the larger of the torsion role's staged history and the Cartesian role's staged pairs (behind its
  // own arrays in the low-register instantiation); sets how many stored pairs the Cartesian role stages
  auto step_lds = [&](CartArgs& cc) -> size_t {
    const int k = L <= 128 ? 0 : 1;
    const size_t arrays = two_per_cu ? CART_ARRAYS_BYTES(L) : 0;
    size_t dyn = L <= 2 * CHAIN_THREADS ? HIST_LDS_BYTES(L) : 0;   // (also covers the Cartesian role's arrays, 100 L bytes, where they are dynamic)
    if (L <= CHAIN_THREADS) {
      const int budget = (two_per_cu ? two_cap : step_dyn_budget(ctx, k, L)) - (int)arrays;
      cc.hist_lds = budget > 0 ? (int)std::min<size_t>(LBM, (size_t)budget / CART_HIST_BYTES(L)) : 0;
      if (const char* e = getenv("TRX2_CART_HIST_LDS")) cc.hist_lds = std::min(cc.hist_lds, std::max(0, atoi(e)));  // A/B and debugging only
      dyn = std::max(dyn, arrays + cc.hist_lds * CART_HIST_BYTES(L));
    }
    return dyn;
  };
  auto enqueue_chunk = [&]() -> int {  // non-zero: a pair launch was refused; nothing was launched after it
    for (int i = 0; i < chunk; i++) {
      const bool samp = pe > 0 && (i % pe) == 0 && (size_t)(3 * prof_used + 2) < ctx->prof_ev.size();
      if (samp) (void)hipEventRecord(ctx->prof_ev[3 * prof_used], ctx->stream);
      if (launch_pair(ctx, B)) return 1;  // (bumps the device-side evaluation counter) never a step kernel on stale records
      if (samp) (void)hipEventRecord(ctx->prof_ev[3 * prof_used + 1], ctx->stream);
      ChainArgs ca = chain_args(ctx, B, MODE_STEP, nruns, max_evals);
      pool_args(ca);
      if (has_cart) {
        // fused launch: workgroups of 256 (512 for 256 < L <= 512) threads, one residue per thread in the Cartesian role
        CartArgs cc = cart_args(ctx, B, nruns, max_evals);
        const dim3 g2(2 * B), b1(CHAIN_THREADS), b2(2 * CHAIN_THREADS);
        const size_t dyn = step_lds(cc);
        if (L <= 128 && two_per_cu) hipLaunchKernelGGL((k_step<1, 128, 128, true>), g2, dim3(128), dyn, ctx->stream, ca, cc);
        else if (L <= 128) hipLaunchKernelGGL((k_step<1, 128, 128>), g2, dim3(128), dyn, ctx->stream, ca, cc);
        else if (two_per_cu) hipLaunchKernelGGL((k_step<1, CHAIN_THREADS, CHAIN_THREADS, true>), g2, b1, dyn, ctx->stream, ca, cc);
        else if (L <= CHAIN_THREADS) hipLaunchKernelGGL((k_step<1, CHAIN_THREADS, CHAIN_THREADS>), g2, b1, dyn, ctx->stream, ca, cc);
        else hipLaunchKernelGGL((k_step<1, 2 * CHAIN_THREADS, 2 * CHAIN_THREADS>), g2, b2, dyn, ctx->stream, ca, cc);
      } else
        launch_chain_args(ctx, B, ca);
      if (samp) { (void)hipEventRecord(ctx->prof_ev[3 * prof_used + 2], ctx->stream); prof_used++; }
    }
    return 0;
  };
  // Shared launches (launch_engine.h): a single-decoy fold -- every feedback iteration of run_inference.py is one -- does not launch
  // for itself; it hands its argument blocks to an engine whose launch pairs step the folds of many contexts at once, and sleeps.
  bool via_engine = false;
  if (want_engine) {
    if (LaunchEngine* E = engine_pick(ctx->device)) {
      EngineJob job;
      job.pa = pair_args(ctx, B);
      job.ca = chain_args(ctx, B, MODE_STEP, nruns, max_evals);
      pool_args(job.ca);
      job.cc = cart_args(ctx, B, nruns, max_evals);
      job.dyn = step_lds(job.cc);
      job.cls = L <= 128 ? 0 : (L <= CHAIN_THREADS ? 1 : 2);
      job.fam_all = ctx->use_orient ? 1 : 0;
      job.wave1 = ctx->pair1_waves == 1 ? 1 : 0;
      job.segc = job.pa.segc ? 1 : 0;
      job.lowreg = two_per_cu ? 1 : 0;
      job.half = (engine_half && job.segc && job.wave1) ? 1 : 0;
      job.bw = ctx->BW; job.B = B; job.n_items = job.pa.n_items; job.done_count = ctx->done_count; job.cap = cap;
      const RowPlan& rp = ctx->plans[(size_t)ctx->plan_cur];
      if (ctx->BW != 1 || rp.epoch != ctx->rows_epoch || rp.pw != 64 || (size_t)rp.ns_max * B * L > ctx->fa_cap) {
        ctx->err = "internal: shared launch shape does not fit the batch buffers or its row plan"; return 1;
      }
      if (!ctx->ev_ready) HIPCHK(hipEventCreateWithFlags(&ctx->ev_ready, hipEventDisableTiming));
      HIPCHK(hipEventRecord(ctx->ev_ready, ctx->stream));
      job.ready = ctx->ev_ready;
      const int erc = engine_run(E, &job);
      if (erc == 1) { ctx->err = job.err; return 1; }
      if (erc == 0) {
        launches = (int)job.launches; slot_launches = (double)job.launches * B;
        *ctx->h_done = job.done;
        via_engine = true;
      }   // (2: the engine was full: this fold launches for itself below)
    }
  }
  // The chunk is a static graph (its only per-evaluation input, the sequence number, lives in device memory): capture it
  // once per (batch shape, protocol length, buffers) and replay it -- 128 launches become one hipGraphLaunch.  Measured on
  // MI355X it changes nothing (the loop is not launch-bound: profiles/README.md), so direct launches stay the default and
  // TRX2_GRAPH=1 opts in.
  // (never on a pool stream another context holds too: its launches from another host thread would land inside the capture)
  const bool no_graph = getenv("TRX2_GRAPH") == nullptr || pool_use_count(ctx->device, ctx->stream) > 1;
  if (!no_graph && !via_engine) {
    const long key[8] = {B, nruns, max_evals, has_cart ? 1 : 0, (long)L * 4096 + N, (long)(seed ^ ((uint64_t)decoy0 << 40) ^ (tors0 ? 1 : 0)), (long)ctx->plan_cur * 128 + ctx->BW, ctx->alloc_epoch};
    if (!ctx->gexec || memcmp(key, ctx->g_key, sizeof key) != 0) {
      if (ctx->gexec) { (void)hipGraphExecDestroy(ctx->gexec); ctx->gexec = nullptr; }
      hipGraph_t graph = nullptr;
      HIPCHK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
      const int refused = enqueue_chunk();
      HIPCHK(hipStreamEndCapture(ctx->stream, &graph));
      if (refused) { if (graph) (void)hipGraphDestroy(graph); return 1; }
      HIPCHK(hipGraphInstantiate(&ctx->gexec, graph, nullptr, nullptr, 0));
      HIPCHK(hipGraphDestroy(graph));
      memcpy(ctx->g_key, key, sizeof key);
    }
  }
  while (!via_engine) {
    if (no_graph) {
      if (enqueue_chunk()) { (void)hipStreamSynchronize(ctx->stream); return 1; }  // ctx->err says which shape was refused
    } else HIPCHK(hipGraphLaunch(ctx->gexec, ctx->stream));
    launches += chunk;
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(ctx->h_done, ctx->done_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < prof_used; k++) {  // the sampled evaluations of this chunk
      float a = 0, b = 0;
      if (hipEventElapsedTime(&a, ctx->prof_ev[3 * k], ctx->prof_ev[3 * k + 1]) == hipSuccess &&
          hipEventElapsedTime(&b, ctx->prof_ev[3 * k + 1], ctx->prof_ev[3 * k + 2]) == hipSuccess) {
        ctx->prof_pair_ms += a; ctx->prof_step_ms += b; ctx->prof_n++; ctx->prof_a.push_back(a); ctx->prof_b.push_back(b);
      }
    }
    prof_used = 0;
    slot_launches += (double)chunk * B;
    if (*ctx->h_done >= B || launches >= cap) break;
    // Tail compaction.  The queue is empty once the first slot has retired; a pair-kernel wave costs the same while any of its
    // 64 decoys is alive and the survivors are spread over all groups, so a three-group launch stays at full length almost to
    // the end.  Whenever the survivors fit into one group fewer, those of the last group move into retired slots of the others
    // (k_compact_*) and the launches shrink by a group, with that shape's split.  The decoys' arithmetic does not depend on the slot; the split (the order in which a
    // residue's records are added) does, so results equal those without compaction up to rounding, not bitwise
    // (trx2_ctx_set_tail_compaction: 0 off, 1 on, 2 on with the split kept -- bitwise equal to off: tests).
    const int live = B - *ctx->h_done;
    const bool drop_group = ctx->BW == 64 && ctx->Bpad > 64 && live <= ctx->Bpad - 64;
    // ... and below one group the group itself halves (32, 16, .. decoys per wave: the decoy-minor coordinates are laid out anew
    // from the decoy-major copy); the sub-lanes of a narrower wave split a decoy's partner residues differently, so this is
    // mode 1 only
    const bool halve = compact_mode == 1 && !drop_group && ctx->Bpad == ctx->BW && ctx->BW >= 2 && live <= ctx->BW / 2;
    if (compact_mode && no_graph && *ctx->h_done > 0 && live > 0 && (drop_group || halve)) {
      const int Bc = drop_group ? ctx->Bpad - 64 : ctx->BW / 2;  // at most 64 moves either way
      CompactArgs C;
      C.B = B; C.Bc = Bc; C.L = L; C.BW = ctx->BW; C.plan = ctx->plan;
      C.st_i = ctx->st_i; C.st_d = ctx->st_d; C.rho = ctx->rho; C.gram = ctx->gram; C.wcur = ctx->wcur; C.slot_id = ctx->slot_id; C.done_count = ctx->done_count;
      C.X = ctx->X; C.G = ctx->G; C.D = ctx->D; C.XT = ctx->XT; C.geom = ctx->geom; C.S = ctx->S; C.Y = ctx->Y; C.P = ctx->P; C.xyzT = ctx->xyzT;
      C.CX = has_cart ? ctx->CX : nullptr; C.CG = ctx->CG; C.CD = ctx->CD; C.CS = ctx->CS; C.CY = ctx->CY;
      hipLaunchKernelGGL(k_compact_plan, dim3(1), dim3(64), 0, ctx->stream, C);
      hipLaunchKernelGGL(k_compact_move, dim3(64), dim3(256), 0, ctx->stream, C);
      B = Bc; ctx->Bpad = Bc;
      if (drop_group) {
        if (compact_mode == 1) {  // the narrower shape's own row plan
          int k;
          if (plan_get(ctx, 1, Bc / 64, &k)) return 1;
          ctx->plan_cur = k;
        }  // mode 2 keeps the plan (a plan made for more groups serves fewer): bitwise equal to the uncompacted fold
      } else {
        ctx->BW = Bc;
        int k;
        if (plan_get(ctx, 64 / Bc, 1, &k)) return 1;
        ctx->plan_cur = k;
        hipLaunchKernelGGL(k_relayout, dim3((unsigned)(((size_t)Bc * L + 255) / 256)), dim3(256), 0, ctx->stream, Bc, L, Bc, (const float4*)ctx->P, ctx->xyzT);
      }
      HIPCHK(hipGetLastError());
    }
  }
  const bool all_retired = *ctx->h_done >= B;
  // results by decoy id
  std::vector<float> tmpx, tmpt;
  std::vector<int> st((size_t)N * 4);
  if (xyz_out) { tmpx.resize(NL * 16); HIPCHK(hipMemcpyAsync(tmpx.data(), ctx->out_xyz, sizeof(float) * NL * 16, hipMemcpyDeviceToHost, ctx->stream)); }
  if (tors_out) { tmpt.resize(NL * 4); HIPCHK(hipMemcpyAsync(tmpt.data(), ctx->out_X, sizeof(float4) * NL, hipMemcpyDeviceToHost, ctx->stream)); }
  if (e_terms) HIPCHK(hipMemcpyAsync(e_terms, ctx->out_e, sizeof(double) * N * TRX2_NTERMS, hipMemcpyDeviceToHost, ctx->stream));
  if (f_final) HIPCHK(hipMemcpyAsync(f_final, ctx->out_f, sizeof(double) * N, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(st.data(), ctx->out_stat, sizeof(int) * N * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (!all_retired) { ctx->err = "trx2_fold_batch: launch cap reached before every decoy had reported"; return 1; }
  if (xyz_out) for (size_t i = 0; i < NL; i++) memcpy(xyz_out + i * 15, tmpx.data() + i * 16, 15 * sizeof(float));
  if (tors_out) for (size_t i = 0; i < NL; i++) memcpy(tors_out + i * 3, tmpt.data() + i * 4, 3 * sizeof(float));
  double evals = 0;
  for (int i = 0; i < N; i++) {
    if (status) status[i] = st[(size_t)i * 4];
    if (n_evals) n_evals[i] = st[(size_t)i * 4 + 1];
    if (n_iters) n_iters[i] = st[(size_t)i * 4 + 2];
    evals += st[(size_t)i * 4 + 1];
  }
  ctx->last_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  ctx->last_launches = launches;
  ctx->last_slot_eff = evals / slot_launches;
  return 0;
}

// process-wide switch of the shared launches (include/trx2fold.h); takes effect for the folds that start after it
extern "C" int trx2_set_shared_launches(int mode) {
  if (mode < -1 || mode > 1) return 1;
  g_engine_mode = mode;
  return 0;
}
extern "C" int trx2_set_shared_launch_halves(int mode) {
  if (mode < -1 || mode > 1) return 1;
  g_engine_half_mode = mode;
  return 0;
}

// out[9] summed over the device's engines: chunks of ENGINE_CHUNK launch pairs enqueued, folds x chunks (their ratio = folds per
// launch), folds completed, seconds the engine threads spent enqueuing, seconds they waited for the GPU; with
// trx2_set_shared_launch_profiling(1): summed milliseconds of the sampled pair / step launches (one launch pair per chunk, HIP events
// on the engine's stream), their number, and the folds they held
extern "C" int trx2_set_shared_launch_profiling(int on) { g_engine_prof = on ? 1 : 0; return 0; }
extern "C" int trx2_shared_launch_stats(int device, double* out) {
  if (!out) return 1;
  for (int i = 0; i < 9; i++) out[i] = 0;
  std::lock_guard<std::mutex> lk(g_engine_mutex);
  auto it = g_engines.find(device);
  if (it == g_engines.end()) return 0;
  for (LaunchEngine* E : it->second) {
    std::lock_guard<std::mutex> l2(E->mu);
    out[0] += E->st_chunks; out[1] += E->st_jobs; out[2] += E->st_done; out[3] += E->st_enqueue_s; out[4] += E->st_wait_s;
    out[5] += E->st_pair_ms; out[6] += E->st_step_ms; out[7] += E->st_prof_n; out[8] += E->st_prof_folds;
  }
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
  if (!k || !ctx->lane_worker || B < TRX2_LANE_MIN_B || !ctx->L)
    return fold_impl(ctx, B, runs, nruns, seed, decoy0, tors0, max_evals, tors_out, xyz_out, e_terms, f_final, status, n_evals, n_iters);
  const int B0 = (B + 1) / 2, B1 = B - B0;
  const size_t L = (size_t)ctx->L;
  auto t0 = std::chrono::steady_clock::now();
  ctx->lane_worker->submit([&]() {
    return fold_impl(k, B1, runs, nruns, seed, decoy0 + (uint32_t)B0, tors0 ? tors0 + (size_t)B0 * L * 3 : nullptr, max_evals,
                     tors_out ? tors_out + (size_t)B0 * L * 3 : nullptr, xyz_out ? xyz_out + (size_t)B0 * L * 15 : nullptr,
                     e_terms ? e_terms + (size_t)B0 * TRX2_NTERMS : nullptr, f_final ? f_final + B0 : nullptr,
                     status ? status + B0 : nullptr, n_evals ? n_evals + B0 : nullptr, n_iters ? n_iters + B0 : nullptr);
  });
  const int rc0 = fold_impl(ctx, B0, runs, nruns, seed, decoy0, tors0, max_evals, tors_out, xyz_out, e_terms, f_final, status, n_evals, n_iters);
  const int rc1 = ctx->lane_worker->wait();   // (its references to this frame end here)
  if (rc1 != 0 && rc0 == 0) ctx->err = "second lane: " + k->err;
  ctx->last_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  ctx->last_launches = ctx->last_launches > k->last_launches ? ctx->last_launches : k->last_launches;
  return rc0 != 0 ? rc0 : rc1;
}

// ---- K7: feedback step.  Host arrays in and out; the device buffers are scratch owned by the context.
static int fb_reserve(trx2_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->fb_cap) return 0;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->fb_buf) (void)hipFree(ctx->fb_buf);
  ctx->fb_buf = nullptr; ctx->fb_cap = 0;
  HIPCHK(hipMalloc(&ctx->fb_buf, bytes));
  ctx->fb_cap = bytes;
  return 0;
}
static size_t al256(size_t n) { return (n + 255) / 256 * 256; }

// bins of one decoy on the device; the four [L][L] int8 arrays stay in the context's scratch (returned pointers)
static int fb_bins_device(trx2_ctx* ctx, int L, const char* seq, const float* xyz, const double* d_edges, int nd, const double* a_edges,
                          int na, const double* p_edges, int np_, double dmax, size_t extra_bytes, signed char* (&bins)[4], char*& extra) {
  const size_t LL = (size_t)L * L;
  const size_t o_xyz = 0, o_gly = al256((size_t)L * 15 * 4), o_ed = o_gly + al256(L), o_bins = o_ed + al256((size_t)(nd + na + np_) * 8);
  const size_t o_extra = o_bins + 4 * al256(LL);
  if (fb_reserve(ctx, o_extra + extra_bytes)) return 1;
  char* base = (char*)ctx->fb_buf;
  std::vector<unsigned char> gly(L);
  for (int i = 0; i < L; i++) gly[i] = seq[i] == 'G';
  std::vector<double> ed((size_t)nd + na + np_);
  memcpy(ed.data(), d_edges, nd * 8); memcpy(ed.data() + nd, a_edges, na * 8); memcpy(ed.data() + nd + na, p_edges, np_ * 8);
  HIPCHK(hipMemcpyAsync(base + o_xyz, xyz, (size_t)L * 15 * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(base + o_gly, gly.data(), L, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(base + o_ed, ed.data(), ed.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));  // the staging vectors above go out of scope
  FbBinsArgs A;
  A.L = L; A.xyz = (const float*)(base + o_xyz); A.gly = (const unsigned char*)(base + o_gly);
  A.d_edges = (const double*)(base + o_ed); A.a_edges = A.d_edges + nd; A.p_edges = A.a_edges + na;
  A.nd = nd; A.na = na; A.np_ = np_; A.dmax2 = dmax * dmax;
  signed char* b0 = (signed char*)(base + o_bins);
  A.jd = b0; A.jo = b0 + al256(LL); A.jt = b0 + 2 * al256(LL); A.jp = b0 + 3 * al256(LL);
  hipLaunchKernelGGL(k_fb_bins, dim3((unsigned)((LL + 255) / 256)), dim3(256), 0, ctx->stream, A);
  HIPCHK(hipGetLastError());
  bins[0] = A.jd; bins[1] = A.jo; bins[2] = A.jt; bins[3] = A.jp;
  extra = base + o_extra;
  return 0;
}
static bool fb_bins_args_ok(int L, const char* seq, const float* xyz, const double* d_edges, int nd, const double* a_edges, int na,
                            const double* p_edges, int np_) {
  return L >= 2 && L <= 4096 && seq && xyz && d_edges && a_edges && p_edges && nd >= 1 && nd <= 64 && na >= 1 && na <= 64 && np_ >= 1 &&
         np_ <= 64 && strnlen(seq, (size_t)L) == (size_t)L;
}

extern "C" int trx2_feedback_bins(trx2_ctx* ctx, int L, const char* seq, const float* xyz, const double* d_edges, int nd,
                                  const double* a_edges, int na, const double* p_edges, int np_, double dmax,
                                  signed char* jd, signed char* jo, signed char* jt, signed char* jp) {
  if (!ctx) return 1;
  if (!fb_bins_args_ok(L, seq, xyz, d_edges, nd, a_edges, na, p_edges, np_) || !jd || !jo || !jt || !jp) {
    ctx->err = "trx2_feedback_bins: bad arguments (need L residues of sequence and coordinates, edge arrays, four outputs)";
    return 1;
  }
  HIPCHK(hipSetDevice(ctx->device));
  const size_t LL = (size_t)L * L;
  signed char* bins[4]; char* extra;
  if (fb_bins_device(ctx, L, seq, xyz, d_edges, nd, a_edges, na, p_edges, np_, dmax, 0, bins, extra)) return 1;
  signed char* out[4] = {jd, jo, jt, jp};
  for (int c = 0; c < 4; c++) HIPCHK(hipMemcpyAsync(out[c], bins[c], LL, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

// One feedback iteration on the RESIDENT distograms of the context (those of the last trx2_set_map, or of the previous
// step): bins of the decoy, the cumulative `tmp` array (from the previous tmp, or from dist the first time), the
// re-weighted channels, and new restraint tables -- what run_inference.py:75-131 does between two folds, with nothing but
// the decoy's coordinates going in and one float coming back.
extern "C" int trx2_feedback_step(trx2_ctx* ctx, const char* seq, const float* xyz, const double* d_edges, int nd,
                                  const double* a_edges, int na, const double* p_edges, int np_, double dmax, const double* w9,
                                  int angle, float* max_tmp_change) {
  if (!ctx) return 1;
  const int L = ctx->L;
  if (!L || !ctx->cur[0] || ctx->borrows_map) { ctx->err = "trx2_feedback_step: set a map first"; return 1; }
  if (ctx->rst_kind == 2) { ctx->err = "trx2_feedback_step: not defined for AlphaFold-style maps (the reference's feedback works on its own 37-bin maps)"; return 1; }
  if (!fb_bins_args_ok(L, seq, xyz, d_edges, nd, a_edges, na, p_edges, np_) || !w9) { ctx->err = "trx2_feedback_step: bad arguments"; return 1; }
  if (angle && !ctx->use_orient) { ctx->err = "trx2_feedback_step: angle channels requested but the map has none"; return 1; }
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->child && ctx->child->stream) HIPCHK(hipStreamSynchronize(ctx->child->stream));  // it reads the tables rebuilt below
  const size_t LL = (size_t)L * L;
  signed char* bins[4]; char* extra;
  if (fb_bins_device(ctx, L, seq, xyz, d_edges, nd, a_edges, na, p_edges, np_, dmax, 256, bins, extra)) return 1;
  unsigned* dmax_bits = (unsigned*)extra;
  HIPCHK(hipMemsetAsync(dmax_bits, 0, 4, ctx->stream));
  const int nb[4] = {TRX2_ND_BINS, TRX2_NO_BINS, TRX2_NO_BINS, TRX2_NP_BINS};
  auto process = [&](const float* in, float* out, int K, const signed char* b, int norm, unsigned* diff) {
    FbProcArgs A;
    A.L = L; A.K = K; A.norm = norm; A.smooth = norm; A.in = in; A.out = out; A.bins = b; A.max_diff_bits = diff;
    memcpy(A.w, w9, sizeof A.w);
    hipLaunchKernelGGL(k_fb_process, dim3((unsigned)((LL + 127) / 128)), dim3(128), 0, ctx->stream, A);
  };
  // tmp first: the first time its base is the dist array BEFORE this step re-weights it (run_inference.py:101-102, R11)
  if (!ctx->tmp_alt) HIPCHK(hipMalloc((void**)&ctx->tmp_alt, LL * nb[0] * 4));
  process(ctx->has_tmp ? ctx->tmp_cur : ctx->cur[0], ctx->tmp_alt, nb[0], bins[0], 0, dmax_bits);
  { float* t = ctx->tmp_cur; ctx->tmp_cur = ctx->tmp_alt; ctx->tmp_alt = t; ctx->has_tmp = true; }
  // channels: dist with jd; with angles omega with jo, theta with jt, phi with jp (= theta on phi's edges)
  const int chan_bins[4] = {0, 1, 2, 3};
  for (int c = 0; c < (angle ? 4 : 1); c++) {
    if (!ctx->alt[c]) HIPCHK(hipMalloc((void**)&ctx->alt[c], LL * nb[c] * 4));
    process(ctx->cur[c], ctx->alt[c], nb[c], bins[chan_bins[c]], 1, nullptr);
    float* t = ctx->cur[c]; ctx->cur[c] = ctx->alt[c]; ctx->alt[c] = t;
  }
  HIPCHK(hipGetLastError());
  unsigned bits = 0;
  HIPCHK(hipMemcpyAsync(&bits, dmax_bits, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (max_tmp_change) memcpy(max_tmp_change, &bits, 4);
  return build_tables(ctx);
}

// current resident distogram: channel 0..3 = dist, omega, theta, phi; 4 = the cumulative tmp array (after a feedback step)
extern "C" int trx2_get_map(trx2_ctx* ctx, int channel, float* out) {
  if (!ctx) return 1;
  const int nb[5] = {TRX2_ND_BINS, TRX2_NO_BINS, TRX2_NO_BINS, TRX2_NP_BINS, TRX2_ND_BINS};
  const float* src = (channel >= 0 && channel < 4) ? ctx->cur[channel] : (channel == 4 && ctx->has_tmp ? ctx->tmp_cur : nullptr);
  if (!ctx->L || !src || !out) { ctx->err = "trx2_get_map: no such resident array"; return 1; }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemcpyAsync(out, src, (size_t)ctx->L * ctx->L * nb[channel] * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int trx2_feedback_process(trx2_ctx* ctx, int L, int K, const float* in, const signed char* bins, const double* w9,
                                     int norm, int smooth, float* out) {
  if (!ctx) return 1;
  if (L < 2 || L > 4096 || K < 8 || K > 40 || !in || !bins || !w9 || !out) {
    ctx->err = "trx2_feedback_process: bad arguments (8 <= K <= 40 bins per pair)";
    return 1;
  }
  HIPCHK(hipSetDevice(ctx->device));
  const size_t LL = (size_t)L * L, nb = LL * K * 4;
  if (fb_reserve(ctx, 2 * al256(nb) + al256(LL))) return 1;
  char* base = (char*)ctx->fb_buf;
  FbProcArgs A;
  A.L = L; A.K = K; A.norm = norm; A.smooth = smooth;
  A.in = (const float*)base; A.out = (float*)(base + al256(nb)); A.bins = (const signed char*)(base + 2 * al256(nb));
  A.max_diff_bits = nullptr;
  memcpy(A.w, w9, sizeof A.w);
  HIPCHK(hipMemcpyAsync(base, in, nb, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(base + 2 * al256(nb), bins, LL, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_fb_process, dim3((unsigned)((LL + 127) / 128)), dim3(128), 0, ctx->stream, A);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, A.out, nb, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

// GloCon matrix of n decoys (utils_trX2dy/utils.py:543-569); host arrays in and out
extern "C" int trx2_glocon_matrix(trx2_ctx* ctx, int n, int L, const char* seqs, const float* xyz, double dmax, double* out) {
  if (!ctx) return 1;
  if (n < 1 || n > 4096 || L < 2 || L > 4096 || !seqs || !xyz || !out || strnlen(seqs, (size_t)n * L) != (size_t)n * L) {
    ctx->err = "trx2_glocon_matrix: need n decoys of L residues (sequence letters n*L, coordinates [n][L][5][3]) and an [n][n] output";
    return 1;
  }
  HIPCHK(hipSetDevice(ctx->device));
  const size_t LL = (size_t)L * L, nL = (size_t)n * L;
  const size_t o_xyz = 0, o_gly = al256(nL * 15 * 4), o_d6 = o_gly + al256(nL), o_out = o_d6 + al256((size_t)n * LL * 8);
  if (fb_reserve(ctx, o_out + al256((size_t)n * n * 8))) return 1;
  char* base = (char*)ctx->fb_buf;
  std::vector<unsigned char> gly(nL);
  for (size_t i = 0; i < nL; i++) gly[i] = seqs[i] == 'G';
  HIPCHK(hipMemcpyAsync(base + o_xyz, xyz, nL * 15 * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(base + o_gly, gly.data(), nL, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(base + o_out, 0, (size_t)n * n * 8, ctx->stream));
  GloArgs A;
  A.n = n; A.L = L; A.xyz = (const float*)(base + o_xyz); A.gly = (const unsigned char*)(base + o_gly); A.dmax2 = dmax * dmax;
  A.d6 = (double*)(base + o_d6); A.out = (double*)(base + o_out);
  hipLaunchKernelGGL(k_glocon_dist, dim3((unsigned)(((size_t)n * LL + 255) / 256)), dim3(256), 0, ctx->stream, A);
  hipLaunchKernelGGL(k_glocon_pairs, dim3((unsigned)(((size_t)n * n + 63) / 64)), dim3(64), 0, ctx->stream, A);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, A.out, (size_t)n * n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

// reliability scores of n decoys (the reference ranks its initial decoys by them, run_inference.py:60-73)
extern "C" int trx2_reliability_scores(trx2_ctx* ctx, int n, int L, const float* xyz, int* counts) {
  if (!ctx) return 1;
  if (n < 1 || L < 1 || L > 4096 || !xyz || !counts || (long)n * L > (1L << 28)) { ctx->err = "trx2_reliability_scores: need xyz[n][L][5][3] and counts[n][2]"; return 1; }
  HIPCHK(hipSetDevice(ctx->device));
  const size_t nx = (size_t)n * L * 15 * 4, nc = (size_t)n * 2 * 4;
  if (fb_reserve(ctx, al256(nx) + al256(nc))) return 1;
  char* base = (char*)ctx->fb_buf;
  HIPCHK(hipMemcpyAsync(base, xyz, nx, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(base + al256(nx), 0, nc, ctx->stream));
  hipLaunchKernelGGL(k_reliability, dim3((unsigned)(((size_t)n * L + 255) / 256)), dim3(256), 0, ctx->stream, n, L, (const float*)base, (int*)(base + al256(nx)));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(counts, base + al256(nx), nc, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

// C-alpha RMSD and TM-score of every pair between two sets of aligned structures
extern "C" int trx2_superpose_matrix(trx2_ctx* ctx, int n, int m, int L, const float* xa, const float* xb, double l_norm, double* rmsd,
                                     double* tm) {
  if (!ctx) return 1;
  if (n < 1 || m < 0 || L < 3 || L > 1024 || !xa || (!rmsd && !tm) || (long)n * (xb ? m : n) > (1L << 24)) {
    ctx->err = "trx2_superpose_matrix: need n x m structures of 3 <= L <= 1024 aligned residues (xa[n][L][3], xb[m][L][3] or NULL) and an output";
    return 1;
  }
  const bool sym = xb == nullptr;
  if (sym) m = n;
  {  // non-finite coordinates have no superposition (and would keep the TM-score search from ending): refuse them here, at the ABI
    bool finite = true;
    for (size_t k = 0; k < (size_t)n * L * 3 && finite; k++) finite = std::isfinite(xa[k]);
    for (size_t k = 0; xb && k < (size_t)m * L * 3 && finite; k++) finite = std::isfinite(xb[k]);
    if (!finite) { ctx->err = "trx2_superpose_matrix: non-finite coordinates"; return 1; }
  }
  HIPCHK(hipSetDevice(ctx->device));
  if (l_norm <= 0) l_norm = L;
  // seed fragments of the TM-score program's search (evaluate.tm_score): lengths L, L/2, .. (at most six, >= 4), then 4
  std::vector<int2> seeds;
  {
    std::vector<int> lens;
    for (int f = L; f >= 4 && lens.size() < 6; f /= 2) lens.push_back(f);
    if (!lens.empty() && lens.back() > 4) lens.push_back(4);
    for (int lf : lens)
      for (int st = 0; st + lf <= L; st++) seeds.push_back(int2{st, lf});
  }
  SupArgs A;
  A.n = n; A.m = m; A.L = L; A.nseed = (int)seeds.size(); A.symmetric = sym;
  A.npair = sym ? (long)n * (n + 1) / 2 : (long)n * m;
  A.lnorm = l_norm;
  double d0 = l_norm > 21 ? 1.24 * std::pow(l_norm - 15.0, 1.0 / 3.0) - 1.8 : 0.5;
  if (d0 < 0.5) d0 = 0.5;
  A.d0 = d0; A.d0_search = d0 < 4.5 ? 4.5 : (d0 > 8.0 ? 8.0 : d0);
  const size_t na = (size_t)n * L * 3 * 4, nb = (size_t)m * L * 3 * 4, nm = (size_t)n * m * 8, ns = seeds.size() * sizeof(int2);
  const size_t o_b = al256(na), o_r = o_b + (sym ? 0 : al256(nb)), o_t = o_r + al256(nm), o_s = o_t + al256(nm);
  if (fb_reserve(ctx, o_s + al256(ns + 16))) return 1;
  char* base = (char*)ctx->fb_buf;
  HIPCHK(hipMemcpyAsync(base, xa, na, hipMemcpyHostToDevice, ctx->stream));
  if (!sym) HIPCHK(hipMemcpyAsync(base + o_b, xb, nb, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(base + o_r, 0, 2 * al256(nm), ctx->stream));
  if (!seeds.empty()) HIPCHK(hipMemcpyAsync(base + o_s, seeds.data(), ns, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));  // `seeds` is read by the copy above
  A.xa = (const float*)base; A.xb = sym ? A.xa : (const float*)(base + o_b);
  A.rmsd = (double*)(base + o_r); A.tm_bits = (unsigned long long*)(base + o_t); A.seeds = (const int2*)(base + o_s);
  const long pairs = A.npair;
  if (rmsd) hipLaunchKernelGGL(k_sup_rmsd, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, ctx->stream, A);
  if (tm && A.nseed) {
    const dim3 grid((unsigned)pairs, (unsigned)((A.nseed + 3) / 4));  // pairs <= 2^24 on x; seed blocks (L <= 1024: < 1600) on y
    if (L <= 128) hipLaunchKernelGGL((k_sup_tm<2>), grid, dim3(256), 0, ctx->stream, A);
    else if (L <= 256) hipLaunchKernelGGL((k_sup_tm<4>), grid, dim3(256), 0, ctx->stream, A);
    else if (L <= 512) hipLaunchKernelGGL((k_sup_tm<8>), grid, dim3(256), 0, ctx->stream, A);
    else hipLaunchKernelGGL((k_sup_tm<16>), grid, dim3(256), 0, ctx->stream, A);
  }
  HIPCHK(hipGetLastError());
  if (rmsd) HIPCHK(hipMemcpyAsync(rmsd, A.rmsd, nm, hipMemcpyDeviceToHost, ctx->stream));
  if (tm) HIPCHK(hipMemcpyAsync(tm, A.tm_bits, nm, hipMemcpyDeviceToHost, ctx->stream));  // bit patterns of doubles
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (sym)
    for (int i = 0; i < n; i++)
      for (int j = 0; j < i; j++) {
        if (rmsd) rmsd[(size_t)i * n + j] = rmsd[(size_t)j * n + i];
        if (tm) tm[(size_t)i * n + j] = tm[(size_t)j * n + i];
      }
  return 0;
}

extern "C" int trx2_time_pair_kernel(trx2_ctx* ctx, int B, const float* w, int sep_lo, int sep_hi, int n_rep,
                                     double* ms_avg, double* term_evals) {
  if (!ctx) return 1;
  if (!ctx->L || !ctx->P || B > ctx->Bcap || B > ctx->Bpad || B < 1 || n_rep < 1 || ctx->plan_cur < 0) {
    ctx->err = "trx2_time_pair_kernel: run an eval/fold batch of this size first";
    return 1;
  }
  HIPCHK(hipSetDevice(ctx->device));
  const int L = ctx->L;
  // weights / separation window for every decoy
  std::vector<float> wc((size_t)B * 8, 0.0f);
  for (int i = 0; i < B; i++) {
    float* p = wc.data() + (size_t)i * 8;
    p[0] = w[0]; p[1] = w[1]; p[2] = w[2]; p[3] = w[3]; p[4] = (float)sep_lo; p[5] = (float)sep_hi; p[6] = 1.0f; p[7] = w[7];
  }
  HIPCHK(hipMemcpyAsync(ctx->wcur, wc.data(), wc.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  // The decoy-minor copy is rebuilt from the decoy-major one in the FULL layout of this batch: a fold that ended under tail
  // compaction leaves xyzT in the narrow layout of its last survivors; P is layout-independent.
  hipLaunchKernelGGL(k_relayout, dim3((unsigned)(((size_t)B * L + 255) / 256)), dim3(256), 0, ctx->stream, B, L, ctx->BW, (const float4*)ctx->P, ctx->xyzT);
  if (launch_pair(ctx, B)) return 1;  // warm
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, ctx->stream));
  for (int i = 0; i < n_rep; i++)
    if (launch_pair(ctx, B)) return 1;
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

extern "C" int trx2_ctx_set_tail_compaction(trx2_ctx* ctx, int mode) {
  if (!ctx) return 1;
  if (mode < 0 || mode > 2) { ctx->err = "trx2_ctx_set_tail_compaction: 0 (off), 1 (on) or 2 (on, pair-kernel split kept)"; return 1; }
  ctx->compact = mode;
  if (ctx->child) ctx->child->compact = mode;
  return 0;
}

extern "C" int trx2_ctx_set_single_decoy_waves(trx2_ctx* ctx, int waves) {
  if (!ctx) return 1;
  if (waves != 1 && waves != 4) { ctx->err = "trx2_ctx_set_single_decoy_waves: 1 or 4"; return 1; }
  ctx->pair1_waves = waves;
  if (ctx->child) ctx->child->pair1_waves = waves;
  return 0;
}
extern "C" int trx2_ctx_set_pool(trx2_ctx* ctx, int slots) {
  if (!ctx) return 1;
  if (slots < 0 || slots > 4096) { ctx->err = "trx2_ctx_set_pool: 0 (one slot per decoy) .. 4096 slots"; return 1; }
  ctx->pool = slots;
  if (ctx->child) ctx->child->pool = slots;
  return 0;
}
extern "C" int trx2_last_fold_slot_efficiency(trx2_ctx* ctx, double* eff) {
  if (!ctx || !eff) return 1;
  *eff = ctx->last_slot_eff;
  return 0;
}
extern "C" int trx2_ctx_set_profiling(trx2_ctx* ctx, int every) {
  if (!ctx) return 1;
  if (every < 0 || every > 64) { ctx->err = "trx2_ctx_set_profiling: every in 0..64"; return 1; }
  ctx->prof_every = every;
  if (ctx->child) ctx->child->prof_every = 0;  // the second lane is never sampled: its events would time overlapped kernels
  return 0;
}
extern "C" int trx2_last_fold_kernel_times(trx2_ctx* ctx, double* pair_ms_avg, double* step_ms_avg, int* n_samples) {
  if (!ctx) return 1;
  // An event pair spans whatever happens between its two records: when the host thread is descheduled between recording an event and
  // launching the kernel behind it while the stream has run dry, the sample holds the stall (seen: one 15-ms sample in a fold of 300,
  // which tripled the average).  Samples beyond four medians of their kernel are dropped; n_samples counts the kept ones.
  auto robust = [](const std::vector<float>& v, const std::vector<float>& w, double& ma, double& mb) -> int {
    if (v.empty()) { ma = mb = 0.0; return 0; }
    std::vector<float> sa(v), sb(w);
    std::nth_element(sa.begin(), sa.begin() + sa.size() / 2, sa.end());
    std::nth_element(sb.begin(), sb.begin() + sb.size() / 2, sb.end());
    const float la = 4.0f * sa[sa.size() / 2], lb = 4.0f * sb[sb.size() / 2];
    double a = 0, b = 0; int n = 0;
    for (size_t i = 0; i < v.size(); i++) if (v[i] <= la && w[i] <= lb) { a += v[i]; b += w[i]; n++; }
    ma = n ? a / n : 0.0; mb = n ? b / n : 0.0;
    return n;
  };
  double ma, mb;
  const int n = robust(ctx->prof_a, ctx->prof_b, ma, mb);
  if (pair_ms_avg) *pair_ms_avg = ma;
  if (step_ms_avg) *step_ms_avg = mb;
  if (n_samples) *n_samples = n;
  return 0;
}
extern "C" int trx2_ctx_info(const trx2_ctx* ctx, int key, double* value) {
  if (!ctx || !value) return 1;
  switch (key) {
    case TRX2_INFO_GROUP_WIDTH: *value = ctx->BW; return 0;
    case TRX2_INFO_SLAB_BYTES: *value = (ctx->plan_cur >= 0 ? ctx->plans[(size_t)ctx->plan_cur].ns_avg : 1.0) * PR_REC * 4; return 0;
    case TRX2_INFO_PAIR_WGS: *value = (ctx->plan_cur >= 0 ? (double)ctx->plans[(size_t)ctx->plan_cur].n_items : 0.0) * (ctx->Bpad / ctx->BW); return 0;
    case TRX2_INFO_CART_STAGED:
      *value = (ctx->L >= 1 && ctx->L <= CHAIN_THREADS) ? (double)std::min<size_t>(LBM, (size_t)step_dyn_budget(ctx, ctx->L <= 128 ? 0 : 1, ctx->L) / CART_HIST_BYTES(ctx->L)) : 0.0;
      return 0;
    case TRX2_INFO_LBFGS_M: *value = LBM; return 0;
    case TRX2_INFO_L: *value = ctx->L; return 0;
    default: return 1;
  }
}

extern "C" int trx2_last_fold_stats(trx2_ctx* ctx, double* seconds, int* n_launches) {
  if (!ctx) return 1;
  if (seconds) *seconds = ctx->last_seconds;
  if (n_launches) *n_launches = ctx->last_launches;
  return 0;
}

#ifdef TRX2_SELFCHECK
// checking build only (libtrx2fold_check.so): one-sum energy totals compared with the nine-sum totals inside the step kernels
extern "C" int trx2_debug_selfcheck(unsigned long long* out6, int reset) {
  if (hipMemcpyFromSymbol(out6, HIP_SYMBOL(g_selfcheck), sizeof(unsigned long long) * 6) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[6] = {0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_selfcheck), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
#endif
#ifdef TRX2_DBG
extern "C" int trx2_debug_linesearch(double* out /* [256][12] */) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(double) * 256 * 12) != hipSuccess;
}
#endif
#ifdef TRX2_STAMP
extern "C" int trx2_debug_stamps(unsigned long long* out32) {  // the pair kernel's phase stamps of its last launch (tools/stamp_pair.py)
  return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 32) != hipSuccess;
}
extern "C" int trx2_debug_chain_stamps(unsigned long long* out32, int reset) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_cstamp), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_cstamp), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
#endif
