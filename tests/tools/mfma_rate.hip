// cycles per MFMA for the K=32 and K=16 forms of the 16x16 f16 / bf16 instructions (one wave per SIMD, 4 accumulators)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  h8 a8, b8v;
  h4 a4, b4;
  b8 c8, d8;
  s4 c4, d4;
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(threadIdx.x * 0.001f + i); b8v[i] = (_Float16)(i * 0.5f); c8[i] = (__bf16)(threadIdx.x * 0.001f + i); d8[i] = (__bf16)(i * 0.5f); }
  for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8v[i]; c4[i] = (short)(threadIdx.x + i); d4[i] = (short)i; }
  long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8v, acc[j], 0, 0, 0);
      if (KIND == 1) acc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[j], 0, 0, 0);
      if (KIND == 2) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c8, d8, acc[j], 0, 0, 0);
      if (KIND == 3) acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(c4, d4, acc[j], 0, 0, 0);
    }
  }
  long t1 = clock64();
  float s = 0;
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0) / (4.f * iters);
}
template <int KIND>
static void run(const char* name, float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms, cyc;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&cyc, out + (1 << 20), 4, hipMemcpyDeviceToHost);
  printf("%-28s %8.3f ms  %6.1f ns per MFMA per wave  clock64 ticks per MFMA %.2f\n", name, ms, ms * 1e6 / (4.0 * iters), cyc);
}
int main() {
  float* out;
  hipMalloc(&out, ((1 << 20) + 16) * 4);
  run<0>("16x16x32 f16", out);
  run<1>("16x16x16 f16", out);
  run<2>("16x16x32 bf16", out);
  run<3>("16x16x16 bf16_1k", out);
  return 0;
}
