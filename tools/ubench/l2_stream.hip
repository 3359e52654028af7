// Microbenchmark: how fast does ONE CU stream a 512 KB matrix that sits in L2 (the decode's cell-kernel stream), by the number of
// 16-byte requests a lane keeps in flight?  One workgroup per CU (dynamic LDS forces that), thread = (4 columns, K group) as in
// k_dec_persist: a wave-instruction reads 1 KB of one row.  Rolling window of D requests per lane, trivial consumption.
// Prints bytes per clock per CU for D = 4..32, 8 and 16 waves per CU, 16 and 256 workgroups.
// Build: hipcc --offload-arch=gfx950 -O3 l2_stream.hip -o l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int NT, int D>
__global__ __launch_bounds__(NT) void k(const float* __restrict__ Wm, float* out, long long* cyc, int passes) {
  extern __shared__ float dsm[];
  constexpr int KG = NT / 128, ROWS = 256 / KG;
  const int tid = threadIdx.x, c4 = tid & 127, kg = tid >> 7;
  const float* wc = Wm + (size_t)(kg * ROWS) * 512 + 4 * c4;
  f2 a0 = f2{0.f, 0.f}, a1 = f2{0.f, 0.f};
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int p = 0; p < passes; ++p) {
    const float* w = wc;
    asm volatile("" : "+v"(w));
    float4 wr[D];
#pragma unroll
    for (int i = 0; i < D; ++i) wr[i] = *reinterpret_cast<const float4*>(w + (size_t)i * 512);
#pragma unroll 1
    for (int k0 = 0; k0 < ROWS - D; k0 += D) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
        a0 += f2{wr[i].x, wr[i].y}; a1 += f2{wr[i].z, wr[i].w};
        __builtin_amdgcn_sched_barrier(0);
        wr[i] = *reinterpret_cast<const float4*>(w + (size_t)(k0 + D + i) * 512);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) { a0 += f2{wr[i].x, wr[i].y}; a1 += f2{wr[i].z, wr[i].w}; }
    __syncthreads();
  }
  long long t1 = __builtin_readcyclecounter();
  out[(size_t)blockIdx.x * NT + tid] = a0.x + a0.y + a1.x + a1.y + dsm[tid];
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}

template <int NT, int D>
void run(const float* Wm, float* out, long long* cyc, int grid) {
  const int passes = 200;
  hipFuncSetAttribute((const void*)k<NT, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  k<NT, D><<<grid, NT, 100 * 1024>>>(Wm, out, cyc, 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<NT, D><<<grid, NT, 100 * 1024>>>(Wm, out, cyc, passes);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double bytes = 512.0 * 1024 * passes;
  // c counts SHADER-clock cycles of workgroup 0 (round 3 printed bytes / (c x 24), taking the counter for a 100 MHz one: a column 24 times too
  // small; cycles per pass and the wall rate were right -- 512 KB / 8,544 cycles = 61 B/clk, 145.6 GB/s per CU = 61 B x 2.37 GHz)
  printf("waves/CU %2d  in flight/lane %2d  workgroups %3d: %.1f B/clk/CU (%lld shader-clock cycles per 512 KB pass)  %.1f GB/s/CU wall\n",
         NT / 64, D, grid, bytes / (double)c, c / passes, bytes / (ms * 1e-3) / 1e9);
}

int main() {
  float *Wm, *out; long long* cyc;
  hipMalloc(&Wm, 512 * 1024); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 64);
  hipMemset(Wm, 0, 512 * 1024);
  for (int grid : {16, 256}) {
    run<512, 4>(Wm, out, cyc, grid); run<512, 8>(Wm, out, cyc, grid); run<512, 16>(Wm, out, cyc, grid); run<512, 32>(Wm, out, cyc, grid);
    run<1024, 4>(Wm, out, cyc, grid); run<1024, 8>(Wm, out, cyc, grid); run<1024, 16>(Wm, out, cyc, grid);
  }
  return 0;
}
