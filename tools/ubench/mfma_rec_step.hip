// Microbenchmark: could the LSTM recurrence h.U run on the 16-bit matrix pipe with split operands?  One 512-thread workgroup per CU,
// 2 chunks per workgroup (the C3 shape): z^T [512 gate columns x 16 (2 used)] = U^T . h^T as v_mfma_f32_16x16x32_f16, U^T resident as A
// fragments (wave w owns the 16-column tiles w, w+8, w+16, w+24 = the four gates of units 16w..16w+15: 4 tiles x 4 k-steps x 2 parts =
// 128 VGPRs), h^T through LDS as B fragments (two f16 parts), 3 part products -> 48 MFMAs per wave per step; then the cell update
// in the 8 lanes per wave that hold valid columns, h back to LDS as f16 parts, one barrier.  Prints cycles per step and the clock.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_rec_step.hip -o mfma_rec_step
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(float* out, const float* Uin, long long* cyc, int steps) {
  __shared__ __align__(16) _Float16 hb[2][2][16 * 2 * 8];          // [buffer][part][(k-block 16) x (chunk 2) x 8]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l16 = lane & 15, q = lane >> 4;
  float4 ua[4][4][2];
  for (int m = 0; m < 4; ++m) for (int ks = 0; ks < 4; ++ks) for (int p = 0; p < 2; ++p) {
    h8 v; for (int j = 0; j < 8; ++j) v[j] = (_Float16)(Uin[(tid * 37 + m * 11 + ks * 5 + p * 3 + j) & 4095] * (p ? 0.001f : 1.f));
    ua[m][ks][p] = __builtin_bit_cast(float4, v);
  }
  for (int i = tid; i < 2 * 2 * 256; i += 512) (&hb[0][0][0])[i] = (_Float16)0.01f;
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; ++s) {
    const int cur = s & 1;
    f4v acc[4];
    for (int m = 0; m < 4; ++m) acc[m] = f4v{0.f, 0.f, 0.f, 0.f};
    const _Float16* bp = &hb[cur][0][0] + (q * 2 + (l16 & 1)) * 8;   // lane (n = chunk, k-quarter q): 8 consecutive units
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const h8 bh = *reinterpret_cast<const h8*>(bp + ks * 64), bl = *reinterpret_cast<const h8*>(bp + 512 + ks * 64);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const h8 ah = __builtin_bit_cast(h8, ua[m][ks][0]), al = __builtin_bit_cast(h8, ua[m][ks][1]);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[m], 0, 0, 0);
      }
    }
    if (l16 < 2) {                                       // columns 0, 1 = the two chunks; rows 4 q + i = units 16 w + 4 q + i
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float zi = acc[0][i] * 1e-4f, zf = acc[1][i] * 1e-4f, zg = acc[2][i] * 1e-4f, zo = acc[3][i] * 1e-4f;
        const float ig = __builtin_amdgcn_rcpf(1.f + __expf(-zi)), fg = __builtin_amdgcn_rcpf(1.f + __expf(-zf));
        const float gg = 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * zg)) - 1.f, og = __builtin_amdgcn_rcpf(1.f + __expf(-zo));
        c[i] = fg * c[i] + ig * gg;
        const float hh = og * (2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * c[i])) - 1.f);
        const float sv = hh * 16384.f;
        const _Float16 hi = (_Float16)sv, lo = (_Float16)(sv - (float)hi);
        const int u = 16 * w + 4 * q + i;
        _Float16* dst = &hb[cur ^ 1][0][0] + ((u >> 3) * 2 + l16) * 8 + (u & 7);
        dst[0] = hi; dst[512] = lo;
      }
    }
    __syncthreads();
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 512 + tid] = c[0] + c[1] + c[2] + c[3];
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}
int main() {
  float *out, *U; long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&U, 1 << 16); hipMalloc(&cyc, 8);
  float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  hipMemcpy(U, h, sizeof h, hipMemcpyHostToDevice);
  const int steps = 3000;
  for (int grid : {256, 512}) {
    long long hc = 0; float ms = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, out, U, cyc, steps); hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1); }
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("grid %d: %.1f cycles per step, wall %.3f ms for %d steps = %.3f us per step (clock %.2f GHz); 300 steps = %.3f ms\n", grid, (double)hc / steps, ms, steps,
           ms * 1e3 / steps / (grid / 256), (double)hc / (ms * 1e-3) / 1e9 * (grid / 256), ms / steps * 300 / (grid / 256));
  }
  return 0;
}
