// launch_engine.h -- shared launches: many independent single-group folds advance in ONE (pair, step) launch pair -- included by
// trx2fold.hip (host side only; the kernels are k_pair1_multi / k_step_multi / k_gather_done).
//
// Why.  The product of run_inference.py is its ITERATION phase: per chain (target x model) a sequence of single-decoy folds,
// each on the map the previous decoy re-weighted (run_inference.py:97-139).  A single-decoy evaluation is a launch pair of ~24 us
// that occupies a few hundred of the chip's 1024 SIMDs for a fraction of that time, a chain's folds depend on each other, and HIP
// gives a process four hardware queues: four chains on four streams was the ceiling (3.6 x one chain, round 3).  The chains of a
// batch job (run_inference.py:339-348: `for name in names`, two models each) are independent, so here their evaluations SHARE
// launches: a fold that qualifies hands its argument blocks to an engine and sleeps; the engine's host thread launches, for all
// folds it holds, one k_pair1_multi (blockIdx.z = fold) and one k_step_multi (blockIdx.y = fold) per evaluation, and wakes a fold
// when its decoys have reported.  Each fold keeps its own context -- map, tables, row plan, state, buffers -- and its arithmetic
// does not depend on what shares its launches: results are bit-identical to the fold's own launches (tests).
//
// Shape.  One engine = one stream + one host thread; TRX2_ENGINE_STREAMS engines per device (default 2: the pair kernel of one
// engine's folds overlaps the step kernel of the other's, as two lanes of one fold do).  The host loop keeps TWO chunks of
// ENGINE_CHUNK launch pairs in flight: chunk k is enqueued before chunk k-1 is waited for, so the GPU never idles on the host.  A
// fold found finished after chunk k-1 is still named in chunk k (already enqueued; its workgroups leave at once: PH_DONE), so it
// is released when chunk k has completed -- no launch in flight ever refers to buffers whose owner has been woken.  Folds join
// at chunk boundaries; their start-up work (on their own stream) is ordered before the chunk by an event.
//
// Half-evaluation launches (round 5; EngineJob::half, k_half_multi in kernel_step.h).  A step launch of n folds is n workgroups -- a chain
// of dependent phases on n of 256 CUs -- and a pair launch fills the chip; one after the other, the chip does pair work half of the time
// (kernel-trace timeline, profiles/r05_engine_overlap.txt).  The folds of a launch class are therefore cut into halves X and Y that run
// half an evaluation apart inside a chunk: launch 2i carries pair(X) beside step(Y), launch 2i+1 step(X) beside pair(Y) -- one kernel,
// two roles, disjoint folds, each fold still pair -> step -> pair in stream order.  Both halves are in step at the chunk boundaries
// (the first launch of a chunk has no step role, a last one steps Y alone), so joining, leaving and draining work as before.
#pragma once
#include <condition_variable>

#define ENGINE_MAX_JOBS 96
#define ENGINE_CHUNK 16

struct EngineJob {
  // filled by the submitting fold
  PairArgs pa; ChainArgs ca; CartArgs cc;
  int cls = 1;            // step kernel: 0 chains of <= 128 residues (128 threads), 1 <= 256, 2 <= 512
  int fam_all = 1;        // pair kernel: all channels | distances only
  int segc = 0;           // pair kernel: the instantiation with the segment cache
  int wave1 = 0;          // pair kernel: one wave per row (k_pair1_multi) | the four-wave workgroups (k_pair_multi<1>)
  int lowreg = 0;         // step kernel: the fold's Cartesian arguments are laid out for the low-register instantiation (k_step_multi<.., true>)
  int half = 0;           // half-evaluation launches (k_half_multi): wave1, segc, lowreg, one slot, chains of <= 256 residues
  int bw = 1;             // decoys per wave of the pair kernel (one decoy group per fold)
  int B = 1;              // slots
  int n_items = 0;        // workgroups of the fold's row plan
  size_t dyn = 0;         // dynamic LDS its step workgroups need
  const int* done_count = nullptr;  // device: slots retired
  hipEvent_t ready = nullptr;       // recorded on the fold's own stream behind its start-up work
  long cap = 0;           // launch pairs after which the fold is given up
  // engine side
  unsigned long long id = 0;
  long launches = 0;
  int done = 0;           // last observed count of retired slots
  int state = 0;          // 0 queued, 1 active, 2 draining, 3 released
  long drain_chunk = -1;
  std::string err;
};

struct LaunchEngine {
  int device = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::vector<EngineJob*> queued, active, draining;
  unsigned long long next_id = 1;
  bool started = false, broken = false;
  std::string broken_why;
  // device argument arrays (one set: uploads are ordered on the engine's stream) and pinned staging (ring of 3)
  char* d_args = nullptr; char* h_args[3] = {nullptr, nullptr, nullptr};
  int* d_flags = nullptr; int* h_flags[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
  // statistics (trx2_shared_launch_stats): chunks enqueued, sum over chunks of the folds they held, folds completed, seconds the
  // host thread spent enqueuing / waiting for a chunk
  double st_chunks = 0, st_jobs = 0, st_done = 0, st_enqueue_s = 0, st_wait_s = 0;
  // trx2_set_shared_launch_profiling: ONE launch pair of every chunk (the one in its middle: it == ENGINE_CHUNK / 2) bracketed by events (pair | step), summed here with the folds it held
  hipEvent_t pev[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  double st_pair_ms = 0, st_step_ms = 0, st_prof_n = 0, st_prof_folds = 0;
  size_t load() { return queued.size() + active.size(); }
};

static const size_t ENG_OFF_PA = 0;
static const size_t ENG_OFF_CA = ENG_OFF_PA + sizeof(PairArgs) * ENGINE_MAX_JOBS;
static const size_t ENG_OFF_CC = ENG_OFF_CA + sizeof(ChainArgs) * ENGINE_MAX_JOBS;
static const size_t ENG_OFF_DP = ENG_OFF_CC + sizeof(CartArgs) * ENGINE_MAX_JOBS;
static const size_t ENG_ARGS_BYTES = ENG_OFF_DP + sizeof(int*) * ENGINE_MAX_JOBS;

static std::atomic<int> g_engine_prof{0};   // trx2_set_shared_launch_profiling
static std::atomic<int> g_engine_mode{-1};   // -1: engine_enabled()'s rule; 0 / 1: trx2_set_shared_launches
static std::mutex g_engine_mutex;
static std::map<int, std::vector<LaunchEngine*>> g_engines;   // per device; never destroyed (their threads outlive every context)

// Who launches a single-decoy fold.  Forced by trx2_set_shared_launches(0 / 1) or TRX2_SHARED_LAUNCH=0 / 1; otherwise by the number
// of contexts alive in the process (second lanes not counted): up to four chains fold fastest launching for themselves, each on one
// of the library's four streams (hardware queues); from the fifth on the streams are shared and the engines win.  Measured on
// MI355X, L=150, all channels, microseconds per fold-evaluation with 1 / 2 / 3 / 4 / 6 / 8 / 12 folds in flight
// (profiles/history/runs_r01_r04.sh.txt section r04_run23.sh): own launches 25.3 / 13.2 / 9.3 / 6.9 / 9.7 / 7.3 / 7.1, engines 26.0 / 13.5 / 10.2 / 8.8 / 6.3 / 5.3 / 4.5.
#ifndef TRX2_ENGINE_MIN_CONTEXTS
#define TRX2_ENGINE_MIN_CONTEXTS 5
#endif
static bool engine_enabled(int chains_alive) {
  const int forced = g_engine_mode.load();
  if (forced >= 0) return forced != 0;
  static const int env = [] { const char* e = getenv("TRX2_SHARED_LAUNCH"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
  if (env >= 0) return env != 0;
  return chains_alive >= TRX2_ENGINE_MIN_CONTEXTS;
}

// Half-evaluation launches: forced by trx2_set_shared_launch_halves(0 / 1) or TRX2_ENGINE_HALF=0 / 1 (results are bit-identical either way);
// otherwise by the number of chains alive in the process, as engine_enabled: they pay when a pair launch fills the chip.  Batch mode at L=150,
// 3 / 8 / 16 / 32 targets (6 / 16 / 32 / 64 chains) in flight: 55.8 -> 53.6, 94.5 -> 95.7, 141 -> 162, 170 -> 179 decoys/s (tools/r05_runs.sh run25, run26).
#ifndef TRX2_ENGINE_HALF_MIN_CONTEXTS
#define TRX2_ENGINE_HALF_MIN_CONTEXTS 12
#endif
static std::atomic<int> g_engine_half_mode{-1};
static bool engine_half_enabled(int chains_alive) {
  const int forced = g_engine_half_mode.load();
  if (forced >= 0) return forced != 0;
  static const int env = [] { const char* e = getenv("TRX2_ENGINE_HALF"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
  if (env >= 0) return env != 0;
  return chains_alive >= TRX2_ENGINE_HALF_MIN_CONTEXTS;
}
// ... used by a chunk for a launch class that holds at least this many folds (TRX2_ENGINE_HALF_MIN, A/B timing: 0 / 8 / 12 measured the same
// in batch mode; the classes below fall back to a pair launch and a low-register step launch)
#ifndef TRX2_ENGINE_HALF_MIN_FOLDS
#define TRX2_ENGINE_HALF_MIN_FOLDS 0
#endif
static int engine_half_min_folds() {
  static const int v = [] { const char* e = getenv("TRX2_ENGINE_HALF_MIN"); return e ? atoi(e) : TRX2_ENGINE_HALF_MIN_FOLDS; }();
  return v;
}

template <int FAM, bool SEGC>
static void engine_launch_pair_t(bool wave1, int n_folds, int max_items, hipStream_t st, const PairArgs* a) {
  if (!wave1) {
    hipLaunchKernelGGL((k_pair_multi<FAM, SEGC>), dim3((unsigned)max_items, 1, (unsigned)n_folds), dim3(PAIR_THREADS), 0, st, a);
    return;
  }
  // eight folds or more: every fold's rows on one XCD (kernel_pair.h, k_pair1_multi); TRX2_XCD_GROUPS=0 / 1 forces either layout (A/B timing)
  static const int env_x = getenv("TRX2_XCD_GROUPS") ? atoi(getenv("TRX2_XCD_GROUPS")) : -1;
  const int xg = env_x >= 0 ? (env_x != 0) : (n_folds >= 8);
  const unsigned blocks = xg ? 8u * (unsigned)((n_folds + 7) / 8) * (unsigned)max_items : (unsigned)n_folds * (unsigned)max_items;
  hipLaunchKernelGGL((k_pair1_multi<FAM, SEGC>), dim3(blocks), dim3(64), 0, st, a, n_folds, max_items, xg);
}
static void engine_launch_pair(bool fam_all, bool wave1, bool segc, int n_folds, int max_items, hipStream_t st, const PairArgs* a) {
  if (fam_all) { if (segc) engine_launch_pair_t<FAM_ALL, true>(wave1, n_folds, max_items, st, a); else engine_launch_pair_t<FAM_ALL, false>(wave1, n_folds, max_items, st, a); }
  else { if (segc) engine_launch_pair_t<FAM_DIST | FAM_VDW, true>(wave1, n_folds, max_items, st, a); else engine_launch_pair_t<FAM_DIST | FAM_VDW, false>(wave1, n_folds, max_items, st, a); }
}
static void engine_launch_step(int cls, bool lowreg, dim3 grid, size_t dyn, hipStream_t st, const ChainArgs* a, const CartArgs* c) {
  if (cls == 0 && lowreg) hipLaunchKernelGGL((k_step_multi<1, 128, 128, true>), grid, dim3(128), dyn, st, a, c);
  else if (cls == 0) hipLaunchKernelGGL((k_step_multi<1, 128, 128, false>), grid, dim3(128), dyn, st, a, c);
  else if (cls == 1 && lowreg) hipLaunchKernelGGL((k_step_multi<1, CHAIN_THREADS, CHAIN_THREADS, true>), grid, dim3(CHAIN_THREADS), dyn, st, a, c);
  else if (cls == 1) hipLaunchKernelGGL((k_step_multi<1, CHAIN_THREADS, CHAIN_THREADS, false>), grid, dim3(CHAIN_THREADS), dyn, st, a, c);
  else hipLaunchKernelGGL((k_step_multi<1, 2 * CHAIN_THREADS, 2 * CHAIN_THREADS, false>), grid, dim3(2 * CHAIN_THREADS), dyn, st, a, c);
}
// one half-evaluation launch: the step role for n_step folds (single-slot folds: two workgroups each) beside the pair role for n_pair folds
template <int FAM, int TN>
static void engine_launch_half_t(int n_step, const ChainArgs* ca, const CartArgs* cc, int n_pair, const PairArgs* pa, int max_items, size_t dyn, hipStream_t st) {
  constexpr unsigned SUBW = TN / 64;
  const int xg = n_pair >= 8;   // every fold's rows on one XCD, as in engine_launch_pair_t
  const unsigned units = n_pair <= 0 ? 0u : (xg ? (unsigned)((n_pair + 7) / 8) : (unsigned)n_pair) * (unsigned)max_items;   // one-wave work items (per XCD column with xg)
  const unsigned pair_blocks = (xg ? 8u : 1u) * ((units + SUBW - 1) / SUBW);
  const unsigned blocks = 2u * (unsigned)std::max(n_step, 0) + pair_blocks;
  if (blocks == 0) return;
  hipLaunchKernelGGL((k_half_multi<FAM, true, TN>), dim3(blocks), dim3(TN), n_step > 0 ? dyn : 0, st, ca, cc, std::max(n_step, 0), pa, std::max(n_pair, 0), max_items, xg);
}
static void engine_launch_half(int cls, bool fam_all, int n_step, const ChainArgs* ca, const CartArgs* cc, int n_pair, const PairArgs* pa, int max_items, size_t dyn, hipStream_t st) {
  if (cls == 0) { if (fam_all) engine_launch_half_t<FAM_ALL, 128>(n_step, ca, cc, n_pair, pa, max_items, dyn, st); else engine_launch_half_t<FAM_DIST | FAM_VDW, 128>(n_step, ca, cc, n_pair, pa, max_items, dyn, st); }
  else { if (fam_all) engine_launch_half_t<FAM_ALL, CHAIN_THREADS>(n_step, ca, cc, n_pair, pa, max_items, dyn, st); else engine_launch_half_t<FAM_DIST | FAM_VDW, CHAIN_THREADS>(n_step, ca, cc, n_pair, pa, max_items, dyn, st); }
}

// launch class of a job: folds of one class share a launch pair (same instantiations); a chunk launches every class it holds
static int engine_class(const EngineJob* j) { return ((((j->cls * 2 + j->fam_all) * 2 + j->wave1) * 2 + j->segc) * 2 + j->lowreg) * 2 + j->half; }

static void engine_fail(LaunchEngine* E, const std::string& why) {   // (mu held) a HIP error on the engine's stream: every fold it holds fails loudly
  E->broken = true; E->broken_why = why;
  // Up to two chunks of launch pairs naming these folds' buffers may still be in flight; an owner must not be woken (and free or reuse
  // them) before they have drained.  Best effort: on a broken stream the wait itself may fail, and then nothing more can run on it.
  if (E->stream) (void)hipStreamSynchronize(E->stream);
  for (auto* v : {&E->queued, &E->active, &E->draining})
    for (EngineJob* j : *v) { j->err = "shared launches: " + why; j->state = 3; }
  E->queued.clear(); E->active.clear(); E->draining.clear();
  E->cv_done.notify_all();
}

static void engine_main(LaunchEngine* E) {
  if (hipSetDevice(E->device) != hipSuccess) { std::lock_guard<std::mutex> lk(E->mu); engine_fail(E, "hipSetDevice failed in the engine thread"); return; }
  struct Chunk { std::vector<EngineJob*> jobs; bool valid = false; bool prof = false; int prof_folds = 0; } ch[2];
  std::vector<unsigned long long> uploaded;   // ids of the jobs the device arrays describe, in order
  long k = 0;
  int ring = 0;
  auto chk = [&](hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    std::lock_guard<std::mutex> lk(E->mu);
    engine_fail(E, std::string(what) + ": " + hipGetErrorString(e));
    return false;
  };
  while (true) {
    Chunk& C = ch[k & 1];
    Chunk& Pv = ch[(k + 1) & 1];
    {
      std::unique_lock<std::mutex> lk(E->mu);
      E->cv_work.wait(lk, [&] { return E->broken || !E->queued.empty() || !E->active.empty() || Pv.valid; });
      if (E->broken) return;
      for (EngineJob* j : E->queued) { j->state = 1; E->active.push_back(j); }
      std::vector<EngineJob*> fresh;
      fresh.swap(E->queued);
      C.jobs = E->active;
      lk.unlock();
      for (EngineJob* j : fresh)   // the fold's start-up work (its own stream) comes before the first launch that steps it
        if (!chk(hipStreamWaitEvent(E->stream, j->ready, 0), "hipStreamWaitEvent")) return;
    }
    C.valid = !C.jobs.empty();
    const auto t_enq = std::chrono::steady_clock::now();
    if (C.valid) {
      std::stable_sort(C.jobs.begin(), C.jobs.end(), [](const EngineJob* a, const EngineJob* b) { return engine_class(a) < engine_class(b); });
      const int n = (int)C.jobs.size();
      bool same = uploaded.size() == (size_t)n;
      for (int i = 0; same && i < n; i++) same = uploaded[(size_t)i] == C.jobs[(size_t)i]->id;
      if (!same) {
        char* h = E->h_args[ring];
        ring = (ring + 1) % 3;
        for (int i = 0; i < n; i++) {
          const EngineJob* j = C.jobs[(size_t)i];
          memcpy(h + ENG_OFF_PA + sizeof(PairArgs) * i, &j->pa, sizeof(PairArgs));
          memcpy(h + ENG_OFF_CA + sizeof(ChainArgs) * i, &j->ca, sizeof(ChainArgs));
          memcpy(h + ENG_OFF_CC + sizeof(CartArgs) * i, &j->cc, sizeof(CartArgs));
          memcpy(h + ENG_OFF_DP + sizeof(int*) * i, &j->done_count, sizeof(int*));
        }
        // one copy of the whole block (60 KB at most; the arrays of the previous chunk are read by kernels that come before it on this stream)
        if (!chk(hipMemcpyAsync(E->d_args, h, ENG_ARGS_BYTES, hipMemcpyHostToDevice, E->stream), "hipMemcpyAsync(args)")) return;
        uploaded.resize((size_t)n);
        for (int i = 0; i < n; i++) uploaded[(size_t)i] = C.jobs[(size_t)i]->id;
      }
      const PairArgs* dpa = (const PairArgs*)(E->d_args + ENG_OFF_PA);
      const ChainArgs* dca = (const ChainArgs*)(E->d_args + ENG_OFF_CA);
      const CartArgs* dcc = (const CartArgs*)(E->d_args + ENG_OFF_CC);
      struct Grp { int lo, n, items, maxB, cls, fam, wave1, segc, lowreg, half; size_t dyn; };
      std::vector<Grp> groups;
      for (int i = 0; i < n;) {
        Grp g{i, 0, 0, 0, C.jobs[(size_t)i]->cls, C.jobs[(size_t)i]->fam_all, C.jobs[(size_t)i]->wave1, C.jobs[(size_t)i]->segc, C.jobs[(size_t)i]->lowreg, C.jobs[(size_t)i]->half, 0};
        while (i < n && engine_class(C.jobs[(size_t)i]) == engine_class(C.jobs[(size_t)g.lo])) {
          const EngineJob* j = C.jobs[(size_t)i];
          g.items = std::max(g.items, j->n_items); g.maxB = std::max(g.maxB, j->B); g.dyn = std::max(g.dyn, j->dyn);
          g.n++; i++;
        }
        groups.push_back(g);
      }
      C.prof = g_engine_prof.load() != 0 && groups.size() == 1 && E->pev[k & 1][0] != nullptr;   // (one class: the events bracket one kernel each)
      C.prof_folds = n;
      for (int it = 0; it < ENGINE_CHUNK; it++)
        for (const Grp& g : groups) {
          const bool samp = C.prof && it == ENGINE_CHUNK / 2;
          if (samp) (void)hipEventRecord(E->pev[k & 1][0], E->stream);
          if (g.half && g.n >= engine_half_min_folds()) {
            // halves X = [lo, lo + nx) and Y = the rest, Y half an evaluation behind: pair(X) beside step(Y) of the previous evaluation, then
            // step(X) beside pair(Y).  (The sampled events bracket the two half-evaluation launches instead of a pair | step kernel.)
            const int nx = (g.n + 1) / 2, ny = g.n - nx, y = g.lo + nx;
            engine_launch_half(g.cls, g.fam != 0, it > 0 ? ny : 0, dca + y, dcc + y, nx, dpa + g.lo, g.items, g.dyn, E->stream);
            if (samp) (void)hipEventRecord(E->pev[k & 1][1], E->stream);
            engine_launch_half(g.cls, g.fam != 0, nx, dca + g.lo, dcc + g.lo, ny, dpa + y, g.items, g.dyn, E->stream);
          } else {
            engine_launch_pair(g.fam != 0, g.wave1 != 0, g.segc != 0, g.n, g.items, E->stream, dpa + g.lo);
            if (samp) (void)hipEventRecord(E->pev[k & 1][1], E->stream);
            engine_launch_step(g.cls, g.lowreg != 0, dim3((unsigned)(2 * g.maxB), (unsigned)g.n), g.dyn, E->stream, dca + g.lo, dcc + g.lo);
          }
          if (samp) (void)hipEventRecord(E->pev[k & 1][2], E->stream);
        }
      for (const Grp& g : groups)   // the second halves' last step of the chunk: every fold has now had ENGINE_CHUNK evaluations
        if (g.half && g.n >= engine_half_min_folds() && g.n > 1) { const int nx = (g.n + 1) / 2; engine_launch_half(g.cls, g.fam != 0, g.n - nx, dca + g.lo + nx, dcc + g.lo + nx, 0, dpa, g.items, g.dyn, E->stream); }
      hipLaunchKernelGGL(k_gather_done, dim3(1), dim3(ENGINE_MAX_JOBS), 0, E->stream, n, (const int* const*)(E->d_args + ENG_OFF_DP), E->d_flags);
      if (!chk(hipGetLastError(), "shared launch")) return;
      if (!chk(hipMemcpyAsync(E->h_flags[k & 1], E->d_flags, sizeof(int) * n, hipMemcpyDeviceToHost, E->stream), "hipMemcpyAsync(flags)")) return;
      if (!chk(hipEventRecord(E->ev[k & 1], E->stream), "hipEventRecord")) return;
    }
    const auto t_wait = std::chrono::steady_clock::now();
    {   // (the statistics are read by trx2_shared_launch_stats under the same lock)
      std::lock_guard<std::mutex> lk(E->mu);
      if (C.valid) { E->st_chunks += 1; E->st_jobs += (double)C.jobs.size(); }
      E->st_enqueue_s += std::chrono::duration<double>(t_wait - t_enq).count();
    }
    if (Pv.valid) {
      if (!chk(hipEventSynchronize(E->ev[(k + 1) & 1]), "hipEventSynchronize")) return;
      std::lock_guard<std::mutex> lk(E->mu);
      E->st_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait).count();
      if (Pv.prof) {
        float a = 0, b = 0;
        hipEvent_t* pe = E->pev[(k + 1) & 1];
        if (hipEventElapsedTime(&a, pe[0], pe[1]) == hipSuccess && hipEventElapsedTime(&b, pe[1], pe[2]) == hipSuccess) {
          E->st_pair_ms += a; E->st_step_ms += b; E->st_prof_n += 1; E->st_prof_folds += Pv.prof_folds;
        }
        Pv.prof = false;
      }
      bool woke = false;
      for (size_t i = 0; i < E->draining.size();) {   // named in no launch that is still in flight: the owner may have its buffers back
        EngineJob* j = E->draining[i];
        if (j->drain_chunk <= k - 1) { j->state = 3; E->draining.erase(E->draining.begin() + (long)i); woke = true; }
        else i++;
      }
      const int* fl = E->h_flags[(k + 1) & 1];
      for (size_t i = 0; i < Pv.jobs.size(); i++) {
        EngineJob* j = Pv.jobs[i];
        if (j->state != 1) continue;
        j->launches += ENGINE_CHUNK;
        j->done = fl[i];
        if (j->done >= j->B || j->launches >= j->cap) {
          E->active.erase(std::find(E->active.begin(), E->active.end(), j));
          const bool in_next = C.valid && std::find(C.jobs.begin(), C.jobs.end(), j) != C.jobs.end();
          if (in_next) { j->state = 2; j->drain_chunk = k; E->draining.push_back(j); }
          else { j->state = 3; woke = true; }
          E->st_done += 1;
        }
      }
      Pv.valid = false;
      if (woke) E->cv_done.notify_all();
    }
    k++;
  }
}

static bool engine_start(LaunchEngine* E, int index) {   // (g_engine_mutex held)
  if (hipSetDevice(E->device) != hipSuccess) return false;
  // TRX2_ENGINE_PRIORITY=1 (experiment, round 5): the engines get streams of their own with DIFFERENT priorities -- engine 0 the greatest,
  // the others the least -- so that when both engines' pair kernels are ready the dispatcher serves engine 0's first and engine 1's fills
  // the CUs engine 0's (narrow) step kernel leaves idle: the two engines otherwise lock in phase (pair beside pair, step beside step).
  static const int prio_mode = getenv("TRX2_ENGINE_PRIORITY") ? atoi(getenv("TRX2_ENGINE_PRIORITY")) : 0;
  // TRX2_ENGINE_CUMASK=1 / 2 (experiment, round 5): every engine's stream is confined to its own share of the CUs (1: contiguous blocks of the
  // mask's bits, 2: interleaved bits), so that one engine's pair kernel -- thousands of one-wave workgroups that fill every SIMD they are
  // allowed on -- cannot keep the other engine's step workgroups (256 + 108 registers: they need a nearly empty CU) waiting for a CU to drain.
  static const int cumask_mode = getenv("TRX2_ENGINE_CUMASK") ? atoi(getenv("TRX2_ENGINE_CUMASK")) : 0;
  if (cumask_mode) {
    int n_eng = 2;
    if (const char* e = getenv("TRX2_ENGINE_STREAMS")) n_eng = std::max(1, std::min(3, atoi(e)));
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, E->device) != hipSuccess || n_cu <= 0) return false;
    std::vector<uint32_t> mask((size_t)(n_cu + 31) / 32, 0u);
    for (int c = 0; c < n_cu; c++) {
      const int owner = cumask_mode == 2 ? c % n_eng : (int)((long)c * n_eng / n_cu);
      if (owner == index) mask[(size_t)c / 32] |= 1u << (c % 32);
    }
    if (hipExtStreamCreateWithCUMask(&E->stream, (uint32_t)mask.size(), mask.data()) != hipSuccess) return false;
  } else if (prio_mode) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return false;
    if (hipStreamCreateWithPriority(&E->stream, hipStreamNonBlocking, index == 0 ? greatest : least) != hipSuccess) return false;
  } else {
    // a stream of its own out of the library's four (pool_acquire), marked so that contexts avoid it while others are free
    E->stream = pool_acquire(E->device, nullptr, 1000);
  }
  if (!E->stream) return false;
  bool ok = hipMalloc((void**)&E->d_args, ENG_ARGS_BYTES) == hipSuccess && hipMalloc((void**)&E->d_flags, sizeof(int) * ENGINE_MAX_JOBS) == hipSuccess;
  for (int i = 0; i < 3 && ok; i++) ok = hipHostMalloc((void**)&E->h_args[i], ENG_ARGS_BYTES) == hipSuccess;
  for (int i = 0; i < 2 && ok; i++) ok = hipHostMalloc((void**)&E->h_flags[i], sizeof(int) * ENGINE_MAX_JOBS) == hipSuccess && hipEventCreateWithFlags(&E->ev[i], hipEventDisableTiming) == hipSuccess;
  for (int i = 0; i < 2 && ok; i++) for (int q = 0; q < 3 && ok; q++) ok = hipEventCreate(&E->pev[i][q]) == hipSuccess;
  if (ok) {
    int lds_max = 0;
    if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, E->device) != hipSuccess || lds_max <= 0) lds_max = 65536;
    if (lds_max > 160 * 1024) lds_max = 160 * 1024;
    const void* f[5] = {(const void*)k_step_multi<1, 128, 128, false>, (const void*)k_step_multi<1, CHAIN_THREADS, CHAIN_THREADS, false>,
                        (const void*)k_step_multi<1, 2 * CHAIN_THREADS, 2 * CHAIN_THREADS, false>,
                        (const void*)k_step_multi<1, 128, 128, true>, (const void*)k_step_multi<1, CHAIN_THREADS, CHAIN_THREADS, true>};
    for (int q = 0; q < 5 && ok; q++) {
      hipFuncAttributes fa;
      ok = hipFuncGetAttributes(&fa, f[q]) == hipSuccess &&
           hipFuncSetAttribute(f[q], hipFuncAttributeMaxDynamicSharedMemorySize, lds_max - (int)fa.sharedSizeBytes) == hipSuccess;
    }
  }
  if (!ok) return false;
  std::thread(engine_main, E).detach();
  E->started = true;
  return true;
}

// the least loaded engine of the device (created on first use); nullptr: shared launches are not available -> the fold launches itself
static LaunchEngine* engine_pick(int device) {
  std::lock_guard<std::mutex> lk(g_engine_mutex);
  std::vector<LaunchEngine*>& v = g_engines[device];
  if (v.empty()) {
    int n = 2;
    if (const char* e = getenv("TRX2_ENGINE_STREAMS")) n = std::max(1, std::min(3, atoi(e)));
    for (int i = 0; i < n; i++) {
      LaunchEngine* E = new LaunchEngine();
      E->device = device;
      if (!engine_start(E, i)) { E->broken = true; E->broken_why = "could not start"; }
      v.push_back(E);
    }
  }
  LaunchEngine* best = nullptr;
  size_t best_load = 0;
  for (LaunchEngine* E : v) {
    std::lock_guard<std::mutex> l2(E->mu);
    if (E->broken || !E->started) continue;
    const size_t ld = E->load();
    if (ld >= ENGINE_MAX_JOBS) continue;
    if (!best || ld < best_load) { best = E; best_load = ld; }
  }
  return best;
}

// Hands a prepared fold to an engine and sleeps until its decoys have reported (or its launch cap is reached).  0: done (job->done,
// job->launches are set); 1: error (job->err); 2: the engine is full (other threads filled it since engine_pick looked) -- nothing was
// queued, the fold launches for itself.
static int engine_run(LaunchEngine* E, EngineJob* job) {
  std::unique_lock<std::mutex> lk(E->mu);
  if (E->broken) { job->err = "shared launches: " + E->broken_why; return 1; }
  if (E->load() >= ENGINE_MAX_JOBS) return 2;   // the argument arrays hold ENGINE_MAX_JOBS folds
  job->id = E->next_id++;
  job->state = 0;
  E->queued.push_back(job);
  E->cv_work.notify_one();
  E->cv_done.wait(lk, [&] { return job->state == 3; });
  return job->err.empty() ? 0 : 1;
}
