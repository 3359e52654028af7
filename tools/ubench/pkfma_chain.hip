// Microbenchmark: v_pk_fma_f32 throughput of one SIMD as a function of the number of independent accumulator chains per wave
// (dependent-issue latency) and of waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 pkfma_chain.hip -o pkfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int CH>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters) {
  f2 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f2{0.f, 0.f};
  f2 a = f2{threadIdx.x * 1e-3f, 0.5f}, b = f2{1.0f, threadIdx.x * 2e-3f};
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 64 / CH; ++j)
#pragma unroll
      for (int c = 0; c < CH; ++c) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));
  }
  long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
  for (int c = 0; c < CH; ++c) r += acc[c].x + acc[c].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  const int iters = 4000;
  auto run = [&](const char* name, auto kern, int threads) {
    long long h = 0;
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const int wps = threads / 256;
    printf("%-10s %d wave(s)/SIMD: %.2f cycles per v_pk_fma_f32 per wave, %.2f per SIMD\n", name, wps, (double)h / (iters * 64.0), (double)h / (iters * 64.0 * wps));
  };
  for (int threads : {256, 512, 1024}) {
    run("1 chain", k<1>, threads); run("2 chains", k<2>, threads); run("4 chains", k<4>, threads); run("8 chains", k<8>, threads);
  }
  return 0;
}
