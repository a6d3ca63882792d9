// What a cross-stream hand-off costs the PRODUCING stream on this stack (round 6): a chain of short kernels on stream A with, after
// every kernel, (a) nothing, (b) hipEventRecord(e, A) [+ a wait on stream B], (c) the same event attached to the kernel launch itself
// (hipExtLaunchKernelGGL stopEvent) [+ a wait on stream B], (d) a hipStreamWaitEvent(A, e_from_B) in front of every kernel.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/event_cost scripts/micro/event_cost.hip && /tmp/event_cost
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                   \
      return 1;                                                               \
    }                                                                         \
  } while (0)

__global__ void spin(int us, int* sink) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)us * 100u) __builtin_amdgcn_s_sleep(8);
  if (sink && us < 0) *sink = 1;
}
__global__ void tiny(int* sink) {
  if (sink && threadIdx.x == 1000) *sink = 1;
}

int main() {
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  const int N = 200, REP = 5;
  std::vector<hipEvent_t> ev(N), evb(N);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : evb) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0));
  CK(hipEventCreate(&t1));
  const int us = 10;
  auto run = [&](const char* name, int mode) -> int {
    float best = 1e9f;
    double host_best = 1e9;
    for (int r = 0; r < REP; ++r) {
      CK(hipDeviceSynchronize());
      spin<<<1, 64, 0, A>>>(200, nullptr);  // lets the host run ahead of the GPU: queues are full when the chain starts
      CK(hipEventRecord(t0, A));
      const auto h0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) {
        if (mode == 4 && i >= 2) CK(hipStreamWaitEvent(A, evb[i - 2], 0));  // (d) a wait on what stream B did two kernels ago
        if (mode == 2 || mode == 3) {
          hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, A, nullptr, ev[i], 0, us, (int*)nullptr);
        } else {
          spin<<<256, 256, 0, A>>>(us, nullptr);
        }
        if (mode == 1 || mode == 5 || mode == 4) CK(hipEventRecord(ev[i], A));
        if (mode == 5 || mode == 3 || mode == 4) {  // stream B consumes the event
          CK(hipStreamWaitEvent(B, ev[i], 0));
          tiny<<<1, 64, 0, B>>>(nullptr);
          if (mode == 4) CK(hipEventRecord(evb[i], B));
        }
      }
      const auto h1 = std::chrono::steady_clock::now();
      CK(hipEventRecord(t1, A));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, t0, t1));
      best = ms < best ? ms : best;
      const double hm = std::chrono::duration<double, std::milli>(h1 - h0).count();
      host_best = hm < host_best ? hm : host_best;
    }
    printf("%-78s %7.2f us per kernel on the GPU (kernel itself %d us), host %5.2f us per iteration\n", name, best * 1e3 / N, us, host_best * 1e3 / N);
    return 0;
  };
  if (run("(a) kernels back to back", 0)) return 1;
  if (run("(b) + hipEventRecord after every kernel (nobody waits)", 1)) return 1;
  if (run("(c) event attached to the launch (hipExtLaunchKernelGGL stopEvent), nobody waits", 2)) return 1;
  if (run("(b') hipEventRecord + stream B waits for it and runs a tiny kernel", 5)) return 1;
  if (run("(c') launch-attached event + stream B waits for it and runs a tiny kernel", 3)) return 1;
  if (run("(d) (b') + stream A waits for B's event of two kernels ago in front of every kernel", 4)) return 1;
  return 0;
}
