// Microbenchmark: the gate-sum product with the h operand in SCALAR registers -- lane = gate column, registers = K: a wave holds
// U[128 k][64 columns] in 128 VGPRs, h[k] arrives by s_load (wave-uniform) and enters v_pk_fma_f32 as its scalar source, packed
// over (k, k+1); no LDS operand traffic, no cross-lane reduction.  Compared with tools/ubench/gatesum_rate.hip (h from LDS,
// K split over lanes): is the LDS-fed loop's 95-98 TFLOP/s ceiling set by the FMAs or by the operand delivery?
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize sgpr_gatesum.hip -o sgpr_gatesum
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, const float* __restrict__ Uin, const float* __restrict__ hbuf, long long* cyc, int iters) {
  f2 u[64];                                   // u[i] = (U[2i][col], U[2i+1][col])
  for (int i = 0; i < 64; ++i) u[i] = f2{Uin[threadIdx.x + 64 * i], Uin[threadIdx.x + 64 * i + 7]};
  float acc = 0.f;
  const int wv = threadIdx.x >> 6;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const float* h = hbuf + __builtin_amdgcn_readfirstlane((it + wv) & 7) * 128;     // wave-uniform address: scalar loads
    f2 a0 = f2{acc, 0.f}, a1 = f2{0.f, 0.f}, a2 = f2{0.f, 0.f}, a3 = f2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 64; i += 4) {
      a0 = __builtin_elementwise_fma(f2{h[2 * i + 0], h[2 * i + 1]}, u[i + 0], a0);
      a1 = __builtin_elementwise_fma(f2{h[2 * i + 2], h[2 * i + 3]}, u[i + 1], a1);
      a2 = __builtin_elementwise_fma(f2{h[2 * i + 4], h[2 * i + 5]}, u[i + 2], a2);
      a3 = __builtin_elementwise_fma(f2{h[2 * i + 6], h[2 * i + 7]}, u[i + 3], a3);
    }
    a0 += a1; a2 += a3; a0 += a2;
    acc = a0.x + a0.y;
    asm volatile("" : "+v"(acc));
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * THREADS + threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float *out, *U, *h; long long* cyc; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&U, 1 << 20); hipMalloc(&h, 8 * 128 * 4); hipMalloc(&cyc, 8);
  hipMemset(U, 0, 1 << 20); hipMemset(h, 0, 8 * 128 * 4);
  const int iters = 20000;
  auto run = [&](const char* name, auto kern, int threads) {
    long long hc = 0; float ms = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, U, h, cyc, iters); hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1); }
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    const int wps = threads / 256;
    const double flops = 256.0 * (threads / 64) * iters * 64.0 * 256.0;
    printf("%-28s %d waves/SIMD: %.2f cycles per pk_fma per wave, %.2f per SIMD; wall %.3f ms = %.1f TFLOP/s (clock %.2f GHz)\n", name, wps,
           (double)hc / (iters * 64.0), (double)hc / (iters * 64.0 * wps), ms, flops / (ms * 1e-3) / 1e12, (double)hc / (ms * 1e-3) / 1e9);
  };
  run("h in SGPRs, U 128 VGPRs", k<512>, 512);
  run("h in SGPRs, U 128 VGPRs", k<256>, 256);
  return 0;
}
