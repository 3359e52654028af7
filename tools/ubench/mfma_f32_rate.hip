// Microbenchmark: cycles per v_mfma_f32_16x16x4_f32 / 32x32x2 issued back-to-back by ONE wave per SIMD,
// with 2 or 4 independent accumulator chains.  Build: hipcc --offload-arch=gfx950 -O3 mfma_f32_rate.hip -o mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void k16(float* out, long long* cyc, int iters) {
  f4v acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = f4v{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CHAINS>
__global__ __launch_bounds__(256) void k32(float* out, long long* cyc, int iters) {
  f16v acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  auto run = [&](const char* name, auto kern, int chains, int grid) {
    long long h = 0;
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s grid %3d: %.1f cycles per MFMA (per SIMD, one wave)\n", name, grid, (double)h / (iters * 8.0 * chains));
  };
  run("16x16x4 f32, 2 chains", k16<2>, 2, 1);   run("16x16x4 f32, 2 chains", k16<2>, 2, 256);
  run("16x16x4 f32, 4 chains", k16<4>, 4, 1);   run("16x16x4 f32, 4 chains", k16<4>, 4, 256);
  run("32x32x2 f32, 2 chains", k32<2>, 2, 1);   run("32x32x2 f32, 2 chains", k32<2>, 2, 256);
  return 0;
}
