// Practical HBM ceilings of one MI355X for the access mixes the bandwidth-class kernels have: pure read (reduction),
// pure write, copy (1 read + 1 write), 2 reads + 1 write, and a write with a 384-byte pixel pitch of which 256 bytes are
// written (the up-sample + concat kernel's store pattern).  Grid-stride float4 kernels, 256 threads, 512 MB per stream.
//   hipcc --offload-arch=gfx950 -O3 scripts/hbm_ceiling.hip -o gpurun_out/hbm_ceiling && gpurun_out/hbm_ceiling
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

template <int UN>
__global__ void read_kernel(const float4* __restrict__ a, size_t n, float* __restrict__ out) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * UN) {
    float4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = i + u * stride < n ? a[i + u * stride] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      s.x += v[u].x;
      s.y += v[u].y;
      s.z += v[u].z;
      s.w += v[u].w;
    }
  }
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;  // keep the loads
}

__global__ void write_kernel(float4* __restrict__ a, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = v;
}

template <int UN>
__global__ void copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * UN) {
    float4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (i + u * stride < n) v[u] = a[i + u * stride];
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (i + u * stride < n) b[i + u * stride] = v[u];
  }
}

__global__ void read2_write1_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float4 x = a[i], y = b[i];
    c[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

// pixels of 24 float4 (384 bytes); quads [8, 24) written: 256 of every 384 bytes
__global__ void write_slice_kernel(float4* __restrict__ a, size_t pixels) {
  const int q = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  for (size_t p = (size_t)blockIdx.x * 16 + pl; p < pixels; p += (size_t)gridDim.x * 16) a[p * 24 + 8 + q] = v;
}

template <typename F>
static double time_us(F launch, int reps = 20) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CHECK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / reps;
}

int main() {
  const size_t bytes = (size_t)512 << 20, n = bytes / 16;
  float4 *a, *b, *c;
  float* out;
  CHECK(hipMalloc(&a, bytes));
  CHECK(hipMalloc(&b, bytes));
  CHECK(hipMalloc(&c, bytes));
  CHECK(hipMalloc(&out, 16));
  CHECK(hipMemset(a, 0, bytes));
  CHECK(hipMemset(b, 0, bytes));
  CHECK(hipMemset(c, 0, bytes));
  const int grids[] = {1024, 2048, 4096, 8192, 16384};
  for (int g : grids) {
    const double r1 = time_us([&] { hipLaunchKernelGGL(read_kernel<1>, dim3(g), dim3(256), 0, 0, a, n, out); });
    const double r4 = time_us([&] { hipLaunchKernelGGL(read_kernel<4>, dim3(g), dim3(256), 0, 0, a, n, out); });
    const double w = time_us([&] { hipLaunchKernelGGL(write_kernel, dim3(g), dim3(256), 0, 0, a, n); });
    const double c1 = time_us([&] { hipLaunchKernelGGL(copy_kernel<1>, dim3(g), dim3(256), 0, 0, a, b, n); });
    const double c4 = time_us([&] { hipLaunchKernelGGL(copy_kernel<4>, dim3(g), dim3(256), 0, 0, a, b, n); });
    const double r2w = time_us([&] { hipLaunchKernelGGL(read2_write1_kernel, dim3(g), dim3(256), 0, 0, a, b, c, n); });
    const double ws = time_us([&] { hipLaunchKernelGGL(write_slice_kernel, dim3(g), dim3(256), 0, 0, a, n / 24); });
    const double gb = bytes / 1e3;  // bytes / us -> MB/s; / 1e3 -> GB/s
    printf("grid %5d  read %6.0f  read(x4) %6.0f  write %6.0f  copy %6.0f  copy(x4) %6.0f  2r1w %6.0f  slice-write %6.0f  GB/s\n", g,
           gb / r1, gb / r4, gb / w, 2 * gb / c1, 2 * gb / c4, 3 * gb / r2w, gb * (16.0 / 24.0) / ws);
  }
  return 0;
}
