// mock_hip.h -- TEST ONLY: the dozen HIP calls csrc/launch_engine.h uses, on the CPU, so that its host threading (condition variables,
// detached engine threads, argument arrays for 96 folds, the drain-by-chunk protocol) runs under -fsanitize=thread / address in
// `pytest -m "not gpu"`.  A stream is a FIFO drained by a worker thread (real concurrency with the engine's host
// thread), an event a ticket counter, a kernel launch a closure executed in stream order, device memory is host memory.  The mock kernels
// DEREFERENCE the fold's buffers through the engine's device argument arrays: a fold that is woken -- and frees them -- while a launch that
// names it is still in flight is a heap-use-after-free under ASAN, which is exactly the invariant the engine promises.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <cstdint>
#include <unistd.h>

typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipEventDisableTiming = 2, hipDeviceAttributeMaxSharedMemoryPerBlock = 7,
       hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct hipFuncAttributes { size_t sharedSizeBytes = 0; };

struct MockStream {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  MockStream() { std::thread([this] { run(); }).detach(); }
  void push(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); } cv.notify_one(); }
  void run() {
    for (;;) {
      std::function<void()> f;
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !q.empty(); }); f = std::move(q.front()); q.pop_front(); }
      f();
    }
  }
};
typedef MockStream* hipStream_t;
struct MockEvent { std::mutex mu; std::condition_variable cv; unsigned long long recorded = 0, done = 0; };
typedef MockEvent* hipEvent_t;

static std::atomic<long> g_mock_launches{0};
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "mock"; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); memset(*p, 0, n); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n) { return hipMalloc(p, n); }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new MockEvent(); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, int) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  unsigned long long t;
  { std::lock_guard<std::mutex> lk(e->mu); t = ++e->recorded; }
  s->push([e, t] { std::lock_guard<std::mutex> lk(e->mu); e->done = std::max(e->done, t); e->cv.notify_all(); });   // (notified under the lock: the waiter may destroy the event as soon as it is released)
  return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t e) {
  std::unique_lock<std::mutex> lk(e->mu);
  const unsigned long long t = e->recorded;
  e->cv.wait(lk, [&] { return e->done >= t; });
  return hipSuccess;
}
inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, int) {
  unsigned long long t;
  { std::lock_guard<std::mutex> lk(e->mu); t = e->recorded; }
  s->push([e, t] { std::unique_lock<std::mutex> lk(e->mu); e->cv.wait(lk, [&] { return e->done >= t; }); });
  return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t s) {
  MockEvent e;
  hipEventRecord(&e, s);
  return hipEventSynchronize(&e);
}
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.01f; return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t st) { st->push([=] { memcpy(d, s, n); }); return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 65536; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
enum { hipStreamNonBlocking = 1, hipDeviceAttributeMultiprocessorCount = 9 };
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, unsigned, const unsigned*) { *s = new MockStream(); return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, int, int) { *s = new MockStream(); return hipSuccess; }
inline hipError_t hipFuncGetAttributes(hipFuncAttributes* a, const void*) { *a = hipFuncAttributes(); return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
#define hipLaunchKernelGGL(kernel, grid, block, dyn, stream, ...) \
  do { const dim3 g_ = (grid); (void)(block); (void)(dyn); g_mock_launches++; (stream)->push([=] { kernel(g_, __VA_ARGS__); }); } while (0)

// ---- what launch_engine.h expects from the translation unit that includes it (trx2fold.hip on the GPU side) ----------------------
#define PAIR_THREADS 256
#define CHAIN_THREADS 256
#define FAM_DIST 1
#define FAM_VDW 4
#define FAM_ALL 7
struct PairArgs { const int* evals; int* pairs; };                   // (the real blocks hold ~40 pointers each; the engine only copies them)  pairs: pair launches that named the fold
struct ChainArgs { int* evals; int* done_count; int B; int need; const int* pairs; };
struct CartArgs { int unused; };
inline void mock_pair(const PairArgs& a) { (void)*(volatile const int*)a.evals; ++*a.pairs; }
// one evaluation of a fold: it reports after `need` of them.  ORDER: the step of evaluation e comes after exactly e + 1 pair launches of the
// fold -- a step before its pair launch (or two pair launches without a step between) is what a wrong half-evaluation schedule would do
inline void mock_step(const ChainArgs& a) {
  if (*a.done_count >= a.B) return;               // (retired slots leave at once, as PH_DONE workgroups do)
  if (*a.pairs != *a.evals + 1) { fprintf(stderr, "mock: step of evaluation %d after %d pair launches\n", *a.evals, *a.pairs); _exit(6); }
  if (++*a.evals >= a.need) *a.done_count = a.B;
}
template <int FAM, bool SEGC> void k_pair_multi(dim3 g, const PairArgs* a) { for (unsigned z = 0; z < g.z; z++) mock_pair(a[z]); }
template <int FAM, bool SEGC> void k_pair1_multi(dim3, const PairArgs* a, int n_folds, int, int) { for (int i = 0; i < n_folds; i++) mock_pair(a[i]); }
template <int RPT, int TN, int NT, bool LOWREG> void k_step_multi(dim3 g, const ChainArgs* a, const CartArgs*) { for (unsigned y = 0; y < g.y; y++) mock_step(a[y]); }
// the half-evaluation launch: the two roles touch disjoint folds, in no particular order (here: the pair role first)
template <int FAM, bool SEGC, int TN> void k_half_multi(dim3, const ChainArgs* ca, const CartArgs*, int n_step, const PairArgs* pa, int n_pair, int, int) {
  for (int i = 0; i < n_pair; i++) mock_pair(pa[i]);
  for (int i = 0; i < n_step; i++) mock_step(ca[i]);
}
inline void k_gather_done(dim3, int n, const int* const* dp, int* flags) { for (int i = 0; i < n; i++) flags[i] = *dp[i]; }
inline hipStream_t pool_acquire(int, hipStream_t = nullptr, int = 1) { return new MockStream(); }
