// Cost of a cross-stream dependency (hipEventRecord + hipStreamWaitEvent between consecutive kernels) against in-stream order.
// build: hipcc -O2 --offload-arch=gfx950 -o xstream_latency xstream_latency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_tick(float* x, int spin) {
  float v = x[threadIdx.x];
  for (int i = 0; i < spin; i++) v = v * 1.0001f + 0.5f;
  x[threadIdx.x] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  float *x, *y;
  hipMalloc(&x, 1024); hipMalloc(&y, 1024); hipMemset(x, 0, 1024); hipMemset(y, 0, 1024);
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi);
  const int N = 4000;
  std::vector<hipEvent_t> ev(2 * N);
  for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  for (int spin : {0, 4000}) {   // 0: empty kernels; 4000: ~10 us kernels (the dependency cost hides behind nothing either way)
    for (int rep = 0; rep < 2; rep++) {
      double t0 = now();
      for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_tick, dim3(1), dim3(64), 0, s1, x, spin);
      double t1 = now(); hipStreamSynchronize(s1); double t2 = now();
      if (rep) printf("spin %d same stream: %.2f us per kernel (host %.2f)\n", spin, (t2 - t0) / N * 1e6, (t1 - t0) / N * 1e6);
      t0 = now();
      for (int i = 0; i < N; i++) {
        hipStream_t s = i % 2 ? s2 : s1;
        if (i) hipStreamWaitEvent(s, ev[i - 1], 0);
        hipLaunchKernelGGL(k_tick, dim3(1), dim3(64), 0, s, x, spin);
        hipEventRecord(ev[i], s);
      }
      t1 = now(); hipStreamSynchronize(s1); hipStreamSynchronize(s2); t2 = now();
      if (rep) printf("spin %d alternating streams: %.2f us per kernel (host %.2f)\n", spin, (t2 - t0) / N * 1e6, (t1 - t0) / N * 1e6);
      t0 = now();
      for (int i = 0; i < N; i++) {   // two chains in anti-phase: x on s1,s2,s1..; y on s2,s1,s2..
        hipStream_t sx = i % 2 ? s2 : s1, sy = i % 2 ? s1 : s2;
        if (i) { hipStreamWaitEvent(sx, ev[i - 1], 0); hipStreamWaitEvent(sy, ev[N + i - 1], 0); }
        hipLaunchKernelGGL(k_tick, dim3(1), dim3(64), 0, sx, x, spin);
        hipEventRecord(ev[i], sx);
        hipLaunchKernelGGL(k_tick, dim3(1), dim3(64), 0, sy, y, spin);
        hipEventRecord(ev[N + i], sy);
      }
      t1 = now(); hipStreamSynchronize(s1); hipStreamSynchronize(s2); t2 = now();
      if (rep) printf("spin %d two chains in anti-phase: %.2f us per step of both chains (host %.2f)\n", spin, (t2 - t0) / N * 1e6, (t1 - t0) / N * 1e6);
    }
  }
  return 0;
}
