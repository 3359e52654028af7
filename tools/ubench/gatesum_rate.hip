// Microbenchmark: the recurrence's gate-sum inner loop in isolation -- v_pk_fma_f32 streaming through a register-resident U
// slice (distinct source registers every instruction, op_sel broadcasts of h), h from registers (REG) or from LDS (ds_read_b128,
// broadcast addresses) -- at 2 waves/SIMD with 128 U registers (the 8-wave kernel) and 4 waves/SIMD with 64 (the 16-wave kernel).
// Prints cycles per v_pk_fma_f32 per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize gatesum_rate.hip -o gatesum_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int NU /*f2 pairs per array: 32 (8-wave) or 16*/, bool LDS, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, const float* Uin, long long* cyc, int iters) {
  __shared__ __align__(16) float hs[128];
  f2 u01[NU], u23[NU];
  for (int i = 0; i < NU; ++i) { u01[i] = f2{Uin[threadIdx.x + 64 * i], Uin[threadIdx.x + 64 * i + 7]}; u23[i] = f2{Uin[threadIdx.x + 64 * i + 13], Uin[threadIdx.x + 64 * i + 29]}; }
  if (threadIdx.x < 128) hs[threadIdx.x] = Uin[threadIdx.x] * 0.01f;
  __syncthreads();
  const int kq = threadIdx.x & (128 / NU / 4 * 4 - 1);      // 4 or 8 K slices
  const float4* hp = reinterpret_cast<const float4*>(hs + (NU) * (kq & 3));
  float4 hreg[NU / 4];
  for (int i = 0; i < NU / 4; ++i) hreg[i] = hp[i];
  float acc = 0.f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    f2 a01 = f2{acc, 0.f}, a23 = f2{0.f, 0.f}, b01 = f2{0.f, 0.f}, b23 = f2{0.f, 0.f};
#pragma unroll
    for (int i4 = 0; i4 < NU / 4; ++i4) {
      float4 hv;
      if (LDS) hv = hp[i4]; else hv = hreg[i4];
      a01 = __builtin_elementwise_fma(f2{hv.x, hv.x}, u01[4 * i4 + 0], a01);
      a23 = __builtin_elementwise_fma(f2{hv.x, hv.x}, u23[4 * i4 + 0], a23);
      b01 = __builtin_elementwise_fma(f2{hv.y, hv.y}, u01[4 * i4 + 1], b01);
      b23 = __builtin_elementwise_fma(f2{hv.y, hv.y}, u23[4 * i4 + 1], b23);
      a01 = __builtin_elementwise_fma(f2{hv.z, hv.z}, u01[4 * i4 + 2], a01);
      a23 = __builtin_elementwise_fma(f2{hv.z, hv.z}, u23[4 * i4 + 2], a23);
      b01 = __builtin_elementwise_fma(f2{hv.w, hv.w}, u01[4 * i4 + 3], b01);
      b23 = __builtin_elementwise_fma(f2{hv.w, hv.w}, u23[4 * i4 + 3], b23);
    }
    a01 += b01; a23 += b23;
    acc = (a01.x + a01.y) + (a23.x + a23.y);
    asm volatile("" : "+v"(acc));
    if (LDS) asm volatile("" ::: "memory");
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * THREADS + threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float *out, *U; long long* cyc; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&U, 1 << 20); hipMalloc(&cyc, 8);
  hipMemset(U, 0, 1 << 20);
  const int iters = 20000;
  auto run = [&](const char* name, auto kern, int threads, int nu) {
    long long h = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, U, cyc, iters); hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1); }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const int wps = threads / 256;
    const double npk = 2.0 * nu;     // pk_fma per iteration per wave
    const double flops = 256.0 * (threads / 64) * iters * npk * 256.0;
    printf("%-34s %d waves/SIMD: %.2f cycles per pk_fma per wave, %.2f per SIMD; wall %.3f ms = %.1f TFLOP/s (clock %.2f GHz)\n", name, wps,
           (double)h / (iters * npk), (double)h / (iters * npk * wps), ms, flops / (ms * 1e-3) / 1e12, (double)h / (ms * 1e-3) / 1e9);
  };
  run("U 128 regs, h in registers", k<32, false, 512>, 512, 32);
  run("U 128 regs, h from LDS", k<32, true, 512>, 512, 32);
  run("U 64 regs, h in registers", k<16, false, 1024>, 1024, 16);
  run("U 64 regs, h from LDS", k<16, true, 1024>, 1024, 16);
  run("U 64 regs, h in registers", k<16, false, 512>, 512, 16);
  run("U 64 regs, h in registers", k<16, false, 256>, 256, 16);
  return 0;
}
