// Microbenchmark: do fp32 MFMA (v_mfma_f32_16x16x4_f32) and packed fp32 VALU FMAs (v_pk_fma_f32) of DIFFERENT waves on
// one SIMD overlap, or do they share the FMA lanes?  One 768-thread workgroup per CU like k_lstm_rec_proj: waves 0-7
// (two per SIMD) run NV v_pk_fma_f32 per iteration, waves 8-11 (one per SIMD) NM MFMAs per iteration.
// Prints cycles per iteration for VALU only, MFMA only and both.  Build: hipcc --offload-arch=gfx950 -O3 coexec_f32.hip -o coexec_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int NV, int NM, bool PLAIN>
__global__ __launch_bounds__(768) void k(float* out, long long* cyc, int iters, int mode /*1 valu, 2 mfma, 3 both*/) {
  const int wv = threadIdx.x >> 6;
  float r = 0.f;
  long long t0 = __builtin_readcyclecounter();
  if (wv < 8) {
    if (mode & 1) {
      f2 acc[8];
      for (int c = 0; c < 8; ++c) acc[c] = f2{0.f, 0.f};
      f2 a = f2{threadIdx.x * 1e-3f, 0.5f}, b = f2{1.0f, threadIdx.x * 2e-3f};
      for (int i = 0; i < iters; ++i) {
        if constexpr (PLAIN) {     // the same FMA count as 2 x NV scalar v_fma_f32 (opaque to the SLP packer)
#pragma unroll
          for (int j = 0; j < NV / 8; ++j)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[c].x) : "v"(a.x), "v"(b.x));
              asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[c].y) : "v"(a.y), "v"(b.y));
            }
        } else {
#pragma unroll
          for (int j = 0; j < NV / 8; ++j)
#pragma unroll
            for (int c = 0; c < 8; ++c) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));
        }
        asm volatile("" : "+v"(a));
      }
      for (int c = 0; c < 8; ++c) r += acc[c].x + acc[c].y;
    }
  } else if (mode & 2) {
    f4v acc[4];
    for (int c = 0; c < 4; ++c) acc[c] = f4v{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < NM / 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
      asm volatile("" : "+v"(a));
    }
    for (int c = 0; c < 4; ++c) r += acc[c][0] + acc[c][3];
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 768 + threadIdx.x] = r;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wv] = t1 - t0;
}

int main() {
  float* out; long long* cyc; hipMalloc(&out, 256 * 768 * 4); hipMalloc(&cyc, 12 * 8);
  const int iters = 4000;
  auto run = [&](const char* name, auto kern, int nv, int nm) {
    for (int mode = 1; mode <= 3; ++mode) {
      long long h[12];
      for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(768), 0, 0, out, cyc, iters, mode); hipDeviceSynchronize(); }
      hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
      printf("%-22s mode %d (%s): valu wave %.1f cyc/iter, mfma wave %.1f cyc/iter   [ideal valu 2 waves x %d x 4 = %d, mfma %d x 32 = %d]\n", name, mode,
             mode == 1 ? "valu only" : mode == 2 ? "mfma only" : "both", (double)h[0] / iters, (double)h[8] / iters, nv, 2 * nv * 4, nm, nm * 32);
    }
  };
  run("64 pk_fma | 32 mfma", k<64, 32, false>, 64, 32);
  run("128 plain fma | 32 mfma", k<64, 32, true>, 64, 32);
  run("128 pk_fma | 64 mfma", k<128, 64, false>, 128, 64);
  run("256 plain fma | 64 mfma", k<128, 64, true>, 128, 64);
  return 0;
}
