// K1 -- fused BiLSTM recurrence for gfx950 (replaces the tf.while_loop of LSTMCell steps that
// Encoder.call drives: /root/reference/basecaller.py:48-59, cell math SURVEY.md A.1/A.2).
//
// One 512-thread workgroup (8 waves, 2 per SIMD) owns ONE direction of BT chunks and walks all T
// steps.  The 128x512 fp32 recurrent kernel U (256 KB) does not fit the 160 KB LDS but does fit
// the CU's register file: thread (j = tid>>2, kq = tid&3) keeps U[32kq..32kq+31][g*128+j] for
// the four gates g in 128 VGPRs.  Each step a thread reads its 32-wide slice of h from LDS
// (8 x ds_read_b128, broadcast across the 16 threads of a wave that share kq), does 128 FMAs,
// and a 2-step DPP butterfly over the 4 kq-lanes of a quad completes the 4 gate pre-activations
// of unit j in every lane of the quad.  sigmoid/tanh, the cell update, the write of h to LDS
// (double-buffered -> one barrier per step) and to the layer output happen in the same kernel.
// Layer 0 (F = 1 raw sample or 5 event features per step) folds its input projection in: the
// chunk's whole input window is staged to LDS once (coalesced), and lane kq adds x_t.W[:,kq*128+j]
// + b to its gate before the butterfly.  Layers >= 1 take the pre-projected x.W + b (K0 GEMM)
// one gate value per lane, prefetched one step ahead.
// The reverse direction simply walks t = T-1..0 and, as in the reference (no mask is passed,
// basecaller.py:400,403), starts on the zero padding.
#include "common.h"

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {   // DPP quad_perm, CTRL = sel0 | sel1<<2 | sel2<<4 | sel3<<6
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}

template <int BT, int F>
__global__ __launch_bounds__(512) void k_lstm_rec(RecArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* hs = smem;                       // [2][BT][128]
  float* xs = smem + 2 * BT * RV_U;       // F>0: [BT][T*F]

  const int tid = threadIdx.x;
  const int j = tid >> 2, kq = tid & 3;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * BT;
  const int T = a.T;

  // ---- recurrent kernel slice -> registers.  Slot r of lane kq holds gate (kq + r) & 3, so the
  // quad reduce-scatter below lands gate kq's full sum in lane kq with three DPP adds and no selects.
  // Slots are paired (0,1) / (2,3) for v_pk_fma_f32: plain v_fma_f32 runs at half the fp32 rate.
  f2 u01[32], u23[32];
  {
    const float* Ud = a.U[dir] + (32 * kq) * RV_G + j;
    const int g0 = kq * RV_U, g1 = ((kq + 1) & 3) * RV_U, g2 = ((kq + 2) & 3) * RV_U, g3 = ((kq + 3) & 3) * RV_U;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      u01[i] = f2{Ud[i * RV_G + g0], Ud[i * RV_G + g1]};
      u23[i] = f2{Ud[i * RV_G + g2], Ud[i * RV_G + g3]};
    }
  }
  const float am = kq == 2 ? 2.f : 1.f;      // own gate: tanh (c~) for kq == 2, sigmoid otherwise
  // ---- layer-0 input kernel column for "my" gate (gate index = kq)
  float wx[F > 0 ? F : 1];
  float bx = 0.f;
  if (F > 0) {
#pragma unroll
    for (int f = 0; f < F; ++f) wx[f] = a.W[dir][f * RV_G + kq * RV_U + j];
    bx = a.bias[dir][kq * RV_U + j];
    // stage the chunks' input windows: [BT][T*F], coalesced
    const int per = T * F;
    for (int r = 0; r < BT; ++r) {
      const int b = min(b0 + r, a.B - 1);
      for (int i = tid; i < per; i += 512) xs[r * per + i] = a.x[(size_t)b * per + i];
    }
  }
  // ---- initial state
  float c[BT];
#pragma unroll
  for (int r = 0; r < BT; ++r) {
    const int b = min(b0 + r, a.B - 1);
    c[r] = a.c0[dir] ? a.c0[dir][(size_t)b * RV_U + j] : 0.f;
    if (kq == 0) hs[r * RV_U + j] = a.h0[dir] ? a.h0[dir][(size_t)b * RV_U + j] : 0.f;
  }
  // ---- prefetch of the pre-projected input for step 0
  float xc[BT];
  if (F == 0) {
    const int t0 = dir ? T - 1 : 0;
#pragma unroll
    for (int r = 0; r < BT; ++r) {
      const int b = min(b0 + r, a.B - 1);
      xc[r] = a.x[((size_t)b * T + t0) * (2 * RV_G) + dir * RV_G + kq * RV_U + j];
    }
  }
  __syncthreads();

  float hlast[BT];
  int cur = 0;
  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    float xn[BT];
    if (F == 0) {  // issue next step's loads now; they land while this step computes
      const int tn = dir ? max(t - 1, 0) : min(t + 1, T - 1);
#pragma unroll
      for (int r = 0; r < BT; ++r) {
        const int b = min(b0 + r, a.B - 1);
        xn[r] = a.x[((size_t)b * T + tn) * (2 * RV_G) + dir * RV_G + kq * RV_U + j];
      }
    }
    const float* hc = hs + cur * BT * RV_U;
    float* hn = hs + (cur ^ 1) * BT * RV_U;
    // phase 1: gate pre-activations of every row (packed FMAs + quad reduce-scatter)
    float z[BT];
#pragma unroll
    for (int r = 0; r < BT; ++r) {
      float xv;
      if (F > 0) {
        xv = bx;
#pragma unroll
        for (int f = 0; f < F; ++f) xv = fmaf(xs[r * T * F + t * F + f], wx[f], xv);
      } else {
        xv = xc[r];
      }
      f2 a01 = f2{xv, 0.f}, a23 = f2{0.f, 0.f};
      const float4* hp = reinterpret_cast<const float4*>(hc + r * RV_U + 32 * kq);
#pragma unroll
      for (int i4 = 0; i4 < 8; ++i4) {
        const float4 hv = hp[i4];
        a01 = __builtin_elementwise_fma(f2{hv.x, hv.x}, u01[4 * i4 + 0], a01);
        a23 = __builtin_elementwise_fma(f2{hv.x, hv.x}, u23[4 * i4 + 0], a23);
        a01 = __builtin_elementwise_fma(f2{hv.y, hv.y}, u01[4 * i4 + 1], a01);
        a23 = __builtin_elementwise_fma(f2{hv.y, hv.y}, u23[4 * i4 + 1], a23);
        a01 = __builtin_elementwise_fma(f2{hv.z, hv.z}, u01[4 * i4 + 2], a01);
        a23 = __builtin_elementwise_fma(f2{hv.z, hv.z}, u23[4 * i4 + 2], a23);
        a01 = __builtin_elementwise_fma(f2{hv.w, hv.w}, u01[4 * i4 + 3], a01);
        a23 = __builtin_elementwise_fma(f2{hv.w, hv.w}, u23[4 * i4 + 3], a23);
      }
      // quad reduce-scatter: lane L takes slot 3 of lane L+1, slot 2 of lane L+2, slot 1 of lane L+3
      float zz = a01.x + quad_perm<0x39>(a23.y);
      zz += quad_perm<0x4E>(a23.x);
      zz += quad_perm<0x93>(a01.y);
      z[r] = zz;
    }
    // phase 2: one activation per lane (gate kq), all-gather inside the quad, cell update.
    // All rows sit in one basic block so their dependent chains interleave.
#pragma unroll
    for (int r = 0; r < BT; ++r) {
      const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-am * z[r]));
      const float act = fmaf(sg, am, 1.0f - am);
      const float ig = quad_perm<0x00>(act), fg = quad_perm<0x55>(act);
      const float gg = quad_perm<0xAA>(act), og = quad_perm<0xFF>(act);
      c[r] = fmaf(fg, c[r], ig * gg);
      hlast[r] = og * rv_tanh(c[r]);
    }
    // phase 3: publish h (LDS for the next step, global for the layer output)
    if (kq == 0) {
#pragma unroll
      for (int r = 0; r < BT; ++r) {
        hn[r * RV_U + j] = hlast[r];
        if (b0 + r < a.B)
          a.out[((size_t)(b0 + r) * a.out_T + a.out_t0 + t) * RV_E + dir * RV_U + j] = hlast[r];
      }
    }
    if (F == 0) {
#pragma unroll
      for (int r = 0; r < BT; ++r) xc[r] = xn[r];
    }
    cur ^= 1;
    __syncthreads();
  }
  if (kq == 0) {
#pragma unroll
    for (int r = 0; r < BT; ++r)
      if (b0 + r < a.B) {
        a.hT[dir][(size_t)(b0 + r) * RV_U + j] = hlast[r];
        a.cT[dir][(size_t)(b0 + r) * RV_U + j] = c[r];
      }
  }
}

template <int BT, int F>
void launch_one(const RecArgs& a, hipStream_t s) {
  dim3 grid((a.B + BT - 1) / BT, 2);
  size_t shm = sizeof(float) * (2 * BT * RV_U + (F > 0 ? (size_t)BT * a.T * F : 0));
  hipLaunchKernelGGL((k_lstm_rec<BT, F>), grid, dim3(512), shm, s, a);
}

template <int F>
void launch_f(const RecArgs& a, int bt, hipStream_t s) {
  switch (bt) {
    case 1: launch_one<1, F>(a, s); break;
    case 2: launch_one<2, F>(a, s); break;
    case 4: launch_one<4, F>(a, s); break;
    default: launch_one<8, F>(a, s); break;
  }
}

}  // namespace

void launch_lstm_rec(const RecArgs& a, int F, int rows_per_block, hipStream_t s) {
  if (F == 0) launch_f<0>(a, rows_per_block, s);
  else if (F == 1) launch_f<1>(a, rows_per_block, s);
  else launch_f<5>(a, rows_per_block, s);
}
