// Probe: do two kernels on ONE stream overlap when the second is launched with hipExtAnyOrderLaunch (its AQL packet without the barrier bit)?
// A = 16 workgroups spinning ~40 us, B = 2048 workgroups spinning ~40 us on other data.  Prints the time of 200 (A, B) pairs with and without the flag,
// and checks that an ordered consumer after each pair sees both kernels' writes.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>
#include <vector>
__global__ void spin(unsigned* out, unsigned iters, unsigned tag) {
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)iters) { __builtin_amdgcn_s_sleep(4); }
  if (threadIdx.x == 0) out[blockIdx.x] = tag;
}
__global__ void check(const unsigned* a, int na, const unsigned* b, int nb, unsigned tag, unsigned* bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < na && a[i] != tag) atomicAdd(bad, 1u);
  if (i < nb && b[i] != tag) atomicAdd(bad, 1u);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  const int NA = 16, NB = 2048, REP = 200;
  unsigned *a, *b, *bad;
  CK(hipMalloc(&a, NA * 4)); CK(hipMalloc(&b, NB * 4)); CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const unsigned ticks = 4000;   // wall_clock64 runs at 100 MHz: 40 us
  for (int mode = 0; mode < 3; mode++) {   // 0: plain, 1: B any-order, 2: plain again
    for (int w = 0; w < 20; w++) { hipLaunchKernelGGL(spin, dim3(NA), dim3(256), 0, st, a, ticks, 0u); hipLaunchKernelGGL(spin, dim3(NB), dim3(64), 0, st, b, ticks, 0u); }
    CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 1; r <= REP; r++) {
      hipLaunchKernelGGL(spin, dim3(NA), dim3(256), 0, st, a, ticks, (unsigned)r);
      if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(NB), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, b, ticks, (unsigned)r);
      else hipLaunchKernelGGL(spin, dim3(NB), dim3(64), 0, st, b, ticks, (unsigned)r);
      hipLaunchKernelGGL(check, dim3((NB + 255) / 256), dim3(256), 0, st, a, NA, b, NB, (unsigned)r, bad);
    }
    CK(hipStreamSynchronize(st));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / REP;
    unsigned hb = 0; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("mode %d (%s): %.1f us per (A, B, check) triple; stale reads seen by the ordered consumer: %u\n", mode, mode == 1 ? "B launched any-order" : "plain", us, hb);
  }
  return 0;
}
