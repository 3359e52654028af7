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
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

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
  // (Up = U pre-arranged at load time in this register order, [32 i][512 threads] float4: the 256 KB slice arrives as 32 coalesced
  //  1-KB loads per wave instead of 128 strided dwords per lane -- the prologue of every recurrence launch)
  f2 u01[32], u23[32];
  {
    const float4* Up = reinterpret_cast<const float4*>(a.Up[dir]) + tid;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const float4 v = Up[i * 512];
      u01[i] = f2{v.x, v.y};
      u23[i] = f2{v.z, v.w};
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


// ------------------------------------------------------------------------------------------------
// Layer 0 with the CELL UPDATE ON FOUR EXTRA WAVES.  In k_lstm_rec every one of the 8 waves spends ~40 % of its instruction stream on
// the serial tail of a row (one activation per lane, quad all-gather, a cell update replicated over the quad, tanh, stores): a wave
// issues one VALU instruction per ~5-9 cycles whatever it is, so the step is the sum of two such streams per SIMD.  Here the 8
// U-holding waves only take the packed FMAs and the quad reduce-scatter and leave the four gate sums of every unit in LDS (zs);
// waves 8-11 -- one per SIMD, no U, one unit per lane -- do the cell update of 32 units each (5 exp + 5 rcp + ~12 VALU per row, instead
// of 8 waves x 25 instructions) and publish h.  The BT rows run as two groups half a step apart, so the tail waves work on one
// group while the FMA waves are in the other group's product: two barriers per step.  (One tail wave for all 128 units makes ITS
// SIMD the straggler of every phase: measured slower than the 8-wave kernel.)
#ifdef RV_REC_STAMPS
#define RV_TW_STAMP_INIT() long long ts_busy = 0, ts_wait = 0, ts_a = __builtin_readcyclecounter()
#define RV_TW_BARRIER() do { const long long tb_ = __builtin_readcyclecounter(); __syncthreads(); const long long tc_ = __builtin_readcyclecounter(); \
                             ts_busy += tb_ - ts_a; ts_wait += tc_ - tb_; ts_a = tc_; } while (0)
#define RV_TW_STAMP_OUT() do { if (a.dbg_ts && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0) { \
                                 a.dbg_ts[2 * (threadIdx.x >> 6)] = ts_busy; a.dbg_ts[2 * (threadIdx.x >> 6) + 1] = ts_wait; } } while (0)
#else
#define RV_TW_STAMP_INIT() do {} while (0)
#define RV_TW_BARRIER() __syncthreads()
#define RV_TW_STAMP_OUT() do {} while (0)
#endif

template <int BT, int F>
__global__ __launch_bounds__(768) void k_lstm_rec_tw(RecArgs a) {
  static_assert(BT >= 2 && BT % 2 == 0 && F > 0, "two row groups; layer 0 only");
  constexpr int H = BT / 2;
  constexpr int KH = H >= 2 ? H / 2 : 1;  // rows of a group per tail lane (lanes 0-31 take even rows of the group, 32-63 odd ones)
  extern __shared__ __align__(16) float smem[];
  float* hs = smem;                       // [BT][128]   h per row (single-buffered: written in the phase after the one that read it)
  float* zs = hs + BT * RV_U;             // [BT][512]   gate sums per row, [gate][unit]
  float* xs = zs + BT * RV_G;             // [BT][T*F]

  const int tid = threadIdx.x;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * BT;
  const int T = a.T, per = T * F;
  for (int r = 0; r < BT; ++r) {          // stage the chunks' input windows, coalesced (all 12 waves)
    const int b = min(b0 + r, a.B - 1);
    for (int i = tid; i < per; i += 768) xs[r * per + i] = a.x[(size_t)b * per + i];
  }

  if (tid >= 512) {
    // ---------------- tail waves: wave 8 + w owns units 32 w .. 32 w + 31; lane half = which rows of a group
    const int l = tid & 63, unit = 32 * ((tid - 512) >> 6) + (l & 31), rsel = l >> 5;
    const bool active = H >= 2 || rsel == 0;
    // The VALU arbiter serves waves by priority, then age, and these are the youngest waves of their SIMDs: at equal priority their
    // ~45 instructions per step get the leftover issue slots and finish LAST (stamps: busy 2,265 of 2,540 cycles per step, the
    // whole workgroup waiting at the barrier for them).  They are short: give them the slots when they want them.
    __builtin_amdgcn_s_setprio(3);
    float c[2][KH], hl[2][KH];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int k = 0; k < KH; ++k) {
        const int r = g * H + (H >= 2 ? 2 * k + rsel : 0);
        const int b = min(b0 + r, a.B - 1);
        c[g][k] = a.c0[dir] ? a.c0[dir][(size_t)b * RV_U + unit] : 0.f;
        hl[g][k] = a.h0[dir] ? a.h0[dir][(size_t)b * RV_U + unit] : 0.f;
        if (active) hs[r * RV_U + unit] = hl[g][k];
      }
    auto tail = [&](int g, int t) {
#pragma unroll
      for (int k = 0; k < KH; ++k) {
        const int r = g * H + (H >= 2 ? 2 * k + rsel : 0);
        const float* z = zs + r * RV_G + unit;
        const float cc = fmaf(rv_sigmoid(z[RV_U]), c[g][k], rv_sigmoid(z[0]) * rv_tanh(z[2 * RV_U]));
        const float hh = rv_sigmoid(z[3 * RV_U]) * rv_tanh(cc);
        c[g][k] = cc; hl[g][k] = hh;
        if (active) {
          hs[r * RV_U + unit] = hh;
          if (b0 + r < a.B) a.out[((size_t)(b0 + r) * a.out_T + a.out_t0 + t) * RV_E + dir * RV_U + unit] = hh;
        }
      }
    };
    __syncthreads();                                       // initial state in LDS
    __syncthreads();                                       // FMA waves: z(G0, step 0)
    RV_TW_STAMP_INIT();
    for (int s = 0; s < T; ++s) {
      const int t = dir ? T - 1 - s : s;
      tail(0, t);                                          // phase B(s): G0's tail beside G1's product
      RV_TW_BARRIER();
      tail(1, t);                                          // phase A(s + 1): G1's tail beside G0's next product
      RV_TW_BARRIER();
    }
    RV_TW_STAMP_OUT();
    if (active) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int k = 0; k < KH; ++k) {
          const int r = g * H + (H >= 2 ? 2 * k + rsel : 0);
          if (b0 + r < a.B) {
            a.hT[dir][(size_t)(b0 + r) * RV_U + unit] = hl[g][k];
            a.cT[dir][(size_t)(b0 + r) * RV_U + unit] = c[g][k];
          }
        }
    }
    return;
  }

  // ---------------- FMA waves (thread = unit j x K quarter kq, U slice in registers as in k_lstm_rec)
  const int j = tid >> 2, kq = tid & 3;
  if (tid >= 256) __builtin_amdgcn_s_setprio(1);           // the second-dispatched half loses every arbitration at equal priority
  f2 u01[32], u23[32];
  {
    const float4* Up = reinterpret_cast<const float4*>(a.Up[dir]) + tid;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const float4 v = Up[i * 512];
      u01[i] = f2{v.x, v.y};
      u23[i] = f2{v.z, v.w};
    }
  }
  float wx[F], bx;
#pragma unroll
  for (int f = 0; f < F; ++f) wx[f] = a.W[dir][f * RV_G + kq * RV_U + j];
  bx = a.bias[dir][kq * RV_U + j];
  auto gate_sum = [&](int r, int t) {                       // lane kq leaves gate kq of unit j in zs[r]
    float xv = bx;
#pragma unroll
    for (int f = 0; f < F; ++f) xv = fmaf(xs[r * per + t * F + f], wx[f], xv);
    f2 a01 = f2{xv, 0.f}, a23 = f2{0.f, 0.f};
    const float4* hp = reinterpret_cast<const float4*>(hs + r * RV_U + 32 * kq);
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
    float zz = a01.x + quad_perm<0x39>(a23.y);
    zz += quad_perm<0x4E>(a23.x);
    zz += quad_perm<0x93>(a01.y);
    zs[r * RV_G + kq * RV_U + j] = zz;
    if (H > 1) __builtin_amdgcn_sched_barrier(0);           // one row's h reads and accumulators live at a time
  };
  __syncthreads();                                         // initial state in LDS
  {
    const int t0 = dir ? T - 1 : 0;
#pragma unroll
    for (int r = 0; r < H; ++r) gate_sum(r, t0);
  }
  __syncthreads();
  RV_TW_STAMP_INIT();
  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    const int tn = dir ? max(t - 1, 0) : min(t + 1, T - 1);  // the last step's look-ahead is computed and dropped
#pragma unroll
    for (int r = H; r < BT; ++r) gate_sum(r, t);           // phase B(s): G1's product for step s (h of G1 from phase A(s))
    RV_TW_BARRIER();
#pragma unroll
    for (int r = 0; r < H; ++r) gate_sum(r, tn);           // phase A(s + 1): G0's product for step s + 1 (h of G0 from phase B(s))
    RV_TW_BARRIER();
  }
  RV_TW_STAMP_OUT();
}

// ------------------------------------------------------------------------------------------------
// Layers >= 1 with the input projection computed IN the recurrence kernel, on the matrix pipe.
// The standalone K0 GEMM leaves the VALU idle and the recurrence leaves the MFMA pipe idle; here one
// 768-thread workgroup holds both: waves 0-7 are the recurrence above (same registers, same math),
// waves 8-11 (one per SIMD; 3 x 168 VGPRs fit the 512-register file) project the NEXT block of
// TBK = 16/BT timesteps of the workgroup's own BT chunks, x[16 rows, 256] . W[256, 512] + b, with
// v_mfma_f32_16x16x4_f32, and hand it over through LDS -- no inter-workgroup synchronisation, no
// pre-projected tensor in HBM.  Per block: activations of block n+2 are fetched from global into a
// double-buffered, fragment-ordered LDS image; every projection wave finishes 8 of the 32 column
// tiles, one (tile, K-half) unit per chunk per step, reading its B fragments as one float4 per lane
// per 4 MFMAs from a weight image pre-arranged at load time (Wp: [tile][k-group][lane][4]).
//
// SB = 1 ("split bf16"): the same product on v_mfma_f32_16x16x32_bf16 -- 16x the MACs per clock of the f32 form.  Both
// operands are cut into three bf16 parts (x = xh + xm + xl exactly: 3 x 8 significant bits, round-to-nearest residuals;
// W likewise, once, at load time) and six of the nine part products are summed in f32, smallest first:
// (xh.wl + xl.wh + xm.wm) + (xh.wm + xm.wh) + xh.wh.  Every bf16 x bf16 product is exact in f32; the three dropped products
// are below 2^-23 of |x.w|, the size of the rounding of one f32 product.  48 MFMAs of 16 cycles per 16x16 tile instead of
// 64 of 32, and each holds the SIMD's vector issue for 8 cycles, not 32: the recurrence waves get the SIMD back.
// A image: [part][k-step 8][k-quarter 4][row 16][8 bf16] (+16 B per 256-B block), B image Wsb: [tile][part][k-step][lane][8 bf16].
//
// SB = 2 ("split f16"): two f16 parts per operand and three products, on v_mfma_f32_16x16x32_f16.  x' = 2^14 x (|x| <= 1: x
// is the output of an LSTM layer) and w' = s_n w (s_n = the power of two that brings column n's largest weight into
// [2^13, 2^14]) are cut as v = vh + 2^-11 vl with vh = f16(v), vl = f16(2^11 (v - vh)): 11 + 11 significant bits and the sign
// of the residual, a relative error below 2^-23 -- the operand as f32 holds it, to within its last bit.  The scaling keeps
// both parts of every operand that matters out of the f16 subnormals (threshold 2^-28 of the column's largest weight).
// x'.w' = xh.wh + 2^-11 (xh.wl + xl.wh) [+ 2^-22 xl.wl, dropped]; every f16 x f16 product is exact in f32.  24 MFMAs of
// 16 cycles per tile, and 4 bytes of weight image per element where the bf16 form reads 6: the projection waves stream
// their B fragments from L2 every block, and at 16 rows per fragment that stream (29 B/clk/CU is what L2 delivers) is
// what bounds this role.  Image Wh: [tile][part 2][k-step 8][lane 64][8 f16], then 512 floats 2^-14 / s_n.
template <int BT, int SB>
__global__ __launch_bounds__(768) void k_lstm_rec_proj(RecArgs a) {
  constexpr int TBK = 16 / BT;                 // timesteps per projected block (16 MFMA rows)
  constexpr int XG = 260;                      // k-group stride of the A image (256 + 4: the transposing stores hit 2 banks deep, not 16)
  constexpr int NP = SB == 2 ? 2 : 3;          // parts per operand
  constexpr int XSB = 272, XSP = 32 * XSB, XSBUF = NP * XSP;   // split image: bytes per (k-step, k-quarter) block / per part / per buffer
  extern __shared__ __align__(16) float smem[];
  float* hs = smem;                            // [2][BT][128]
  float* xwb = hs + 2 * BT * RV_U;             // [2][16 rows][512]   projected inputs, row = s_local*BT + r
  float* bsm = xwb + 2 * 16 * RV_G;            // [512] bias of this direction, [512] column scales (SB = 2)
  float* xa = bsm + 2 * RV_G;             // [2][16 k-groups][4 q][16 rows][4 i]  A fragments: lane (q, row) reads one float4

  const int tid = threadIdx.x;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * BT;
  const int T = a.T;
  const int nblk = (T + TBK - 1) / TBK;
  const bool proj = tid >= 512;

  // stage of A rows of block `blk` (walk order) into xa[blk & 1]; 256 threads, 4 float4 each
  const int p = tid - 512;
  auto a_load = [&](int blk, float4* v) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int idx = p + 256 * f, rho = idx >> 6, k4 = idx & 63;
      const int s = min(blk * TBK + rho / BT, T - 1), r = rho % BT;
      const int t = dir ? T - 1 - s : s;
      const int b = min(b0 + r, a.B - 1);
      v[f] = *reinterpret_cast<const float4*>(a.x + ((size_t)b * T + t) * RV_E + 4 * k4);
    }
  };
  auto a_store = [&](int blk, const float4* v) {
    if constexpr (SB) {
      char* dst = reinterpret_cast<char*>(xa) + (blk & 1) * XSBUF;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int idx = p + 256 * f, rho = idx >> 6, k4 = idx & 63;      // k = 4 k4: block k4/2 = 4 (k/32) + (k%32)/8, half k4&1
        char* q0 = dst + (k4 >> 1) * XSB + rho * 16 + (k4 & 1) * 8;
        f2 r0 = f2{v[f].x, v[f].y}, r1 = f2{v[f].z, v[f].w};
        if constexpr (SB == 2) {
          r0 *= 16384.f; r1 *= 16384.f;
#pragma unroll
          for (int part = 0; part < 2; ++part) {
            const h2 c0 = __builtin_convertvector(r0, h2), c1 = __builtin_convertvector(r1, h2);
            *reinterpret_cast<uint2*>(q0 + part * XSP) = uint2{__builtin_bit_cast(unsigned, c0), __builtin_bit_cast(unsigned, c1)};
            r0 = (r0 - __builtin_convertvector(c0, f2)) * 2048.f; r1 = (r1 - __builtin_convertvector(c1, f2)) * 2048.f;   // exact
          }
        } else {
#pragma unroll
          for (int part = 0; part < 3; ++part) {
            const bf2 c0 = __builtin_convertvector(r0, bf2), c1 = __builtin_convertvector(r1, bf2);
            *reinterpret_cast<uint2*>(q0 + part * XSP) = uint2{__builtin_bit_cast(unsigned, c0), __builtin_bit_cast(unsigned, c1)};
            r0 -= __builtin_convertvector(c0, f2); r1 -= __builtin_convertvector(c1, f2);   // exact: the residual fits f32
          }
        }
      }
      return;
    }
    float* dst = xa + (blk & 1) * (16 * XG);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int idx = p + 256 * f, rho = idx >> 6, k4 = idx & 63;
      float* q0 = dst + (k4 >> 2) * XG + rho * 4 + (k4 & 3);            // [g][q][rho][i]: g = k/16, i = (k%16)/4, q = k%4
      q0[0] = v[f].x; q0[64] = v[f].y; q0[128] = v[f].z; q0[192] = v[f].w;   // q = element index; a lane's float4 = its 4 k-steps
    }
  };

  // The two roles run separate loops with the same number of workgroup barriers (T + TBK step barriers after one
  // prologue barrier): s_barrier counts arriving waves, not program counters, and separate loops keep the register
  // allocation of the recurrence (128 VGPRs of U) apart from the projection's fragments.
  if (proj) {
    const int lane = tid & 63, pw = p >> 6;             // projection wave 0..3 owns column tiles 8 pw .. 8 pw + 7
    f4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float4 av[4];                                       // activations of a later block in flight
    a_load(0, av); a_store(0, av);
    if (nblk > 1) { a_load(1, av); a_store(1, av); }
    bsm[p] = a.bias[dir][p]; bsm[p + 256] = a.bias[dir][p + 256];
    if constexpr (SB == 2) {                              // 2^-14 / s_n behind the bias
      const float* cs = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.Wh[dir]) + (size_t)2 * RV_E * RV_G * 2);
      bsm[RV_G + p] = cs[p]; bsm[RV_G + p + 256] = cs[p + 256];
    }
    __syncthreads();
    if constexpr (SB) {
      // unit = (column tile, K half) as below: 4 k-steps x 6 (3) part products = 24 (12) MFMAs; the B parts of the NEXT unit
      // are requested before this unit's MFMAs (4 NP x 16 B per lane), the A parts are read from LDS one k-step at a time.
      f4v accs = {0.f, 0.f, 0.f, 0.f}, accm = {0.f, 0.f, 0.f, 0.f}, acch = {0.f, 0.f, 0.f, 0.f};
      const char* wimg = reinterpret_cast<const char*>(SB == 2 ? a.Wh[dir] : a.Wsb[dir]);
      auto b_issue = [&](int unit, float4* bf) {          // bf[4 part + ks]
        const int nt = 8 * pw + ((unit & 15) >> 1), kh = unit & 1;
        const char* bp = wimg + ((size_t)(nt * NP) * 8 + 4 * kh) * 1024 + lane * 16;
#pragma unroll
        for (int part = 0; part < NP; ++part)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) bf[4 * part + ks] = *reinterpret_cast<const float4*>(bp + (size_t)(part * 8 + ks) * 1024);
      };
      auto proj_unit = [&](int blk, int unit, const float4* bf, float4* bf_next) {
        const int nt = 8 * pw + (unit >> 1), kh = unit & 1;
        b_issue(unit + 1, bf_next);
        const char* ap = reinterpret_cast<const char*>(xa) + (blk & 1) * XSBUF + (16 * kh + (lane >> 4)) * XSB + (lane & 15) * 16;
        if (kh == 0) { accs = f4v{0.f, 0.f, 0.f, 0.f}; accm = accs; acch = accs; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          float4 af[NP];
#pragma unroll
          for (int part = 0; part < NP; ++part) af[part] = *reinterpret_cast<const float4*>(ap + part * XSP + ks * 4 * XSB);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (SB == 2) {
            const h8 ah = __builtin_bit_cast(h8, af[0]), al = __builtin_bit_cast(h8, af[1]);
            const h8 bh = __builtin_bit_cast(h8, bf[ks]), bl = __builtin_bit_cast(h8, bf[4 + ks]);
            accm = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, accm, 0, 0, 0);
            acch = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acch, 0, 0, 0);
            accs = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, accs, 0, 0, 0);
          } else {
            const bf8 ah = __builtin_bit_cast(bf8, af[0]), am = __builtin_bit_cast(bf8, af[1]), al = __builtin_bit_cast(bf8, af[2]);
            const bf8 bh = __builtin_bit_cast(bf8, bf[ks]), bm = __builtin_bit_cast(bf8, bf[4 + ks]), bl = __builtin_bit_cast(bf8, bf[8 + ks]);
            accs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, accs, 0, 0, 0);
            accm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, accm, 0, 0, 0);
            acch = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acch, 0, 0, 0);
            accs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, accs, 0, 0, 0);
            accm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, accm, 0, 0, 0);
            accs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, accs, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (kh == 1) {                                     // tile done: + bias -> LDS (C/D map: col = lane%16, row = 4*(lane/16)+i)
          const int col = 16 * nt + (lane & 15);
          const float bb = bsm[col];
          float* dst = xwb + (blk & 1) * (16 * RV_G) + (4 * (lane >> 4)) * RV_G + col;
          if constexpr (SB == 2) {
            const float cs = bsm[RV_G + col];
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[i * RV_G] = fmaf(fmaf(accs[i] + accm[i], 0x1p-11f, acch[i]), cs, bb);
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[i * RV_G] = ((accs[i] + accm[i]) + acch[i]) + bb;
          }
        }
      };
      float4 bfa[4 * NP], bfb[4 * NP];
      b_issue(0, bfa);
      RV_TW_STAMP_INIT();
      for (int s = -TBK; s < T; ++s) {
        const int blk = s >= 0 ? s / TBK : -1;
        const int sl = s - blk * TBK;
        const int nb = blk + 1;
        if (nb < nblk && a.dbg_role != 1) {
          if (BT == 1) {
            proj_unit(nb, sl, bfa, bfb);
#pragma unroll
            for (int g = 0; g < 4 * NP; ++g) bfa[g] = bfb[g];
          } else {
#pragma unroll
            for (int u = 0; u < BT; u += 2) {
              proj_unit(nb, sl * BT + u, bfa, bfb);
              proj_unit(nb, sl * BT + u + 1, bfb, bfa);
            }
          }
        }
        if (nb >= 1 && nb + 1 < nblk) {
          if (sl == 0) a_load(nb + 1, av);
          if (sl == TBK - 1) a_store(nb + 1, av);
        }
        RV_TW_BARRIER();
      }
      RV_TW_STAMP_OUT();
      return;
    }
    // One projection "unit" = (column tile, K half): 8 k-groups = 32 MFMAs.  A block needs 16 units per wave;
    // BT of them run per step, so block n+1 is complete exactly when the recurrence finishes block n.
    // This wave is alone with its MFMAs on its SIMD, so nothing hides its load latency: the B fragments of the
    // NEXT unit are requested before the MFMAs of the current one (the unit -> address map does not depend on
    // the block), the A fragments are read from LDS in one batch at the start of a unit.
    auto b_issue = [&](int unit, float4* bf) {
      const int nt = 8 * pw + ((unit & 15) >> 1), kh = unit & 1;
      const float* bp = a.Wp[dir] + (((size_t)nt * 16 + 8 * kh) * 64 + lane) * 4;
#pragma unroll
      for (int g = 0; g < 8; ++g) bf[g] = *reinterpret_cast<const float4*>(bp + (size_t)g * 256);
    };
    auto a_issue = [&](int blk, int unit, float4* af) {
      const float* ap = xa + (blk & 1) * (16 * XG) + (8 * (unit & 1)) * XG + lane * 4;   // [g][q = lane/16][rho = lane%16][i]
#pragma unroll
      for (int g = 0; g < 8; ++g) af[g] = *reinterpret_cast<const float4*>(ap + g * XG);
    };
    // unit = request the NEXT unit's B fragments (global), read this unit's A fragments (LDS), then 32 MFMAs on the
    // fragments requested one unit ago.  The scheduling fences keep the requests in front of the MFMAs: vmcnt retires in
    // order, so waiting for this unit's fragments leaves the next unit's 8 loads in flight.
    auto proj_unit = [&](int blk, int unit, const float4* bf, float4* bf_next) {
      const int nt = 8 * pw + (unit >> 1), kh = unit & 1;
      b_issue(unit + 1, bf_next);
      float4 af[8];
      a_issue(blk, unit, af);
      __builtin_amdgcn_sched_barrier(0);
      if (kh == 0) { acc0 = f4v{0.f, 0.f, 0.f, 0.f}; acc1 = f4v{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g].x, bf[g].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g].y, bf[g].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g].z, bf[g].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g].w, bf[g].w, acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kh == 1) {                                     // tile done: + bias -> LDS (C/D map: col = lane%16, row = 4*(lane/16)+i)
        const int col = 16 * nt + (lane & 15);
        const float bb = bsm[col];
        float* dst = xwb + (blk & 1) * (16 * RV_G) + (4 * (lane >> 4)) * RV_G + col;
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i * RV_G] = (acc0[i] + acc1[i]) + bb;
      }
    };
    float4 bfa[8], bfb[8];
    b_issue(0, bfa);
    for (int s = -TBK; s < T; ++s) {
      const int blk = s >= 0 ? s / TBK : -1;            // block the recurrence is in (-1: warm-up while block 0 is projected)
      const int sl = s - blk * TBK;                     // step inside the block, 0..TBK-1
      const int nb = blk + 1;                           // block being projected
      if (nb < nblk && a.dbg_role != 1) {
        if (BT == 1) {                                   // one unit per step: the fragment buffers rotate through registers
          proj_unit(nb, sl, bfa, bfb);
#pragma unroll
          for (int g = 0; g < 8; ++g) bfa[g] = bfb[g];
        } else {
#pragma unroll
          for (int u = 0; u < BT; u += 2) {              // BT is even: units alternate between the two fragment buffers
            proj_unit(nb, sl * BT + u, bfa, bfb);
            proj_unit(nb, sl * BT + u + 1, bfb, bfa);
          }
        }
      }
      // activations of block nb+1: fetched at the first step of this block, stored at its last step (that LDS image
      // was last read while block nb-1 was projected)
      if (nb >= 1 && nb + 1 < nblk) {
        if (sl == 0) a_load(nb + 1, av);
        if (sl == TBK - 1) a_store(nb + 1, av);
      }
      __syncthreads();
    }
    return;
  }

  // ---------------- recurrence role (the loop body is k_lstm_rec<BT, 0>'s, with x.W + b read from LDS)
  const int j = tid >> 2, kq = tid & 3;
  f2 u01[32], u23[32];
  float c[BT], hlast[BT];
  const float am = kq == 2 ? 2.f : 1.f;
  {
    const float4* Up = reinterpret_cast<const float4*>(a.Up[dir]) + tid;      // tid < 512 in this role
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const float4 v = Up[i * 512];
      u01[i] = f2{v.x, v.y};
      u23[i] = f2{v.z, v.w};
    }
#pragma unroll
    for (int r = 0; r < BT; ++r) {
      const int b = min(b0 + r, a.B - 1);
      c[r] = a.c0[dir] ? a.c0[dir][(size_t)b * RV_U + j] : 0.f;
      hlast[r] = 0.f;
      if (kq == 0) hs[r * RV_U + j] = a.h0[dir] ? a.h0[dir][(size_t)b * RV_U + j] : 0.f;
    }
  }
  __syncthreads();
  RV_TW_STAMP_INIT();
  for (int s = -TBK; s < 0; ++s) RV_TW_BARRIER();       // block 0 is being projected
  int cur = 0;
  for (int s = 0; s < T; ++s) {
    if (a.dbg_role == 2) { RV_TW_BARRIER(); continue; }
    const int blk = s / TBK, sl = s - blk * TBK;
    const int t = dir ? T - 1 - s : s;
    const float* hc = hs + cur * BT * RV_U;
    float* hn = hs + (cur ^ 1) * BT * RV_U;
    const float* xrow = xwb + (blk & 1) * (16 * RV_G) + (sl * BT) * RV_G + kq * RV_U + j;
    float z[BT];
#pragma unroll
    for (int r = 0; r < BT; ++r) {
      f2 a01 = f2{xrow[r * RV_G], 0.f}, a23 = f2{0.f, 0.f};
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
      float zz = a01.x + quad_perm<0x39>(a23.y);
      zz += quad_perm<0x4E>(a23.x);
      zz += quad_perm<0x93>(a01.y);
      z[r] = zz;
    }
#pragma unroll
    for (int r = 0; r < BT; ++r) {
      const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-am * z[r]));
      const float act = fmaf(sg, am, 1.0f - am);
      const float ig = quad_perm<0x00>(act), fg = quad_perm<0x55>(act);
      const float gg = quad_perm<0xAA>(act), og = quad_perm<0xFF>(act);
      c[r] = fmaf(fg, c[r], ig * gg);
      hlast[r] = og * rv_tanh(c[r]);
    }
    if (kq == 0) {
#pragma unroll
      for (int r = 0; r < BT; ++r) {
        hn[r * RV_U + j] = hlast[r];
        if (b0 + r < a.B)
          a.out[((size_t)(b0 + r) * a.out_T + a.out_t0 + t) * RV_E + dir * RV_U + j] = hlast[r];
      }
    }
    cur ^= 1;
    RV_TW_BARRIER();
  }
  RV_TW_STAMP_OUT();
  if (kq == 0) {
#pragma unroll
    for (int r = 0; r < BT; ++r)
      if (b0 + r < a.B) {
        a.hT[dir][(size_t)(b0 + r) * RV_U + j] = hlast[r];
        a.cT[dir][(size_t)(b0 + r) * RV_U + j] = c[r];
      }
  }
}

constexpr size_t proj_lds_bytes(int bt, int split) {
  return sizeof(float) * (2 * bt * RV_U + 2 * 16 * RV_G + 2 * RV_G) + (split ? 2 * (split == 2 ? 2 : 3) * 32 * 272 : sizeof(float) * 2 * 16 * 260);
}
template <int BT>
void launch_proj(const RecArgs& a, hipStream_t s) {
  dim3 grid((a.B + BT - 1) / BT, 2);
  if (a.Wh[0]) hipLaunchKernelGGL((k_lstm_rec_proj<BT, 2>), grid, dim3(768), proj_lds_bytes(BT, 2), s, a);
  else if (a.Wsb[0]) hipLaunchKernelGGL((k_lstm_rec_proj<BT, 1>), grid, dim3(768), proj_lds_bytes(BT, 1), s, a);
  else hipLaunchKernelGGL((k_lstm_rec_proj<BT, 0>), grid, dim3(768), proj_lds_bytes(BT, 0), s, a);
}

constexpr size_t REC_LDS_CAP = 160 * 1024;      // gfx950: 160 KB per workgroup (these kernels hold no static LDS)
constexpr size_t one_lds_bytes(int bt, int F, int T) { return sizeof(float) * (2 * (size_t)bt * RV_U + (F > 0 ? (size_t)bt * T * F : 0)); }
constexpr size_t tw_lds_bytes(int bt, int F, int T) { return sizeof(float) * ((size_t)bt * RV_U + (size_t)bt * RV_G + (size_t)bt * T * F); }

template <int BT, int F>
void launch_one(const RecArgs& a, hipStream_t s) {
  dim3 grid((a.B + BT - 1) / BT, 2);
  hipLaunchKernelGGL((k_lstm_rec<BT, F>), grid, dim3(512), one_lds_bytes(BT, F, a.T), s, a);
}

template <int BT, int F>
void launch_tw(const RecArgs& a, hipStream_t s) {
  dim3 grid((a.B + BT - 1) / BT, 2);
  hipLaunchKernelGGL((k_lstm_rec_tw<BT, F>), grid, dim3(768), tw_lds_bytes(BT, F, a.T), s, a);
}

template <int F>
void launch_f(const RecArgs& a, int bt, hipStream_t s) {
  if constexpr (F > 0) {
    // (the tail-wave form stages 12 KB more per 8 rows than the 8-wave form: very long windows fall back to the latter)
    if (a.tail_wave && bt >= 2 && tw_lds_bytes(bt, F, a.T) <= REC_LDS_CAP) {
      switch (bt) {
        case 2: launch_tw<2, F>(a, s); return;
        case 4: launch_tw<4, F>(a, s); return;
        default: launch_tw<8, F>(a, s); return;
      }
    }
  }
  switch (bt) {
    case 1: launch_one<1, F>(a, s); break;
    case 2: launch_one<2, F>(a, s); break;
    case 4: launch_one<4, F>(a, s); break;
    default: launch_one<8, F>(a, s); break;
  }
}

}  // namespace

void launch_lstm_rec_proj(const RecArgs& a, int rows_per_block, hipStream_t s) {
  switch (rows_per_block) {
    case 1: launch_proj<1>(a, s); break;
    case 2: launch_proj<2>(a, s); break;
    case 4: launch_proj<4>(a, s); break;
    default: launch_proj<8>(a, s); break;
  }
}
bool lstm_rec_window_fits(int F, int rows_per_block, int T) {
  return one_lds_bytes(rows_per_block, F, T) <= REC_LDS_CAP;     // launch_f falls back to the 8-wave form when the tail-wave form does not fit
}
hipError_t configure_rec_kernels() {
  const int shm = (int)proj_lds_bytes(8, 1);             // the largest of the three images
  hipError_t first = hipSuccess;
  // layer-0 kernels stage the chunks' whole input windows: opt in to the full 160 KB (the 64 KB default is reached at
  // T_event > 281 / T_raw > 1408 with 8 rows per workgroup)
  for (const void* f : {reinterpret_cast<const void*>(&k_lstm_rec<1, 1>), reinterpret_cast<const void*>(&k_lstm_rec<2, 1>),
                        reinterpret_cast<const void*>(&k_lstm_rec<4, 1>), reinterpret_cast<const void*>(&k_lstm_rec<8, 1>),
                        reinterpret_cast<const void*>(&k_lstm_rec<1, 5>), reinterpret_cast<const void*>(&k_lstm_rec<2, 5>),
                        reinterpret_cast<const void*>(&k_lstm_rec<4, 5>), reinterpret_cast<const void*>(&k_lstm_rec<8, 5>),
                        reinterpret_cast<const void*>(&k_lstm_rec_tw<2, 1>), reinterpret_cast<const void*>(&k_lstm_rec_tw<4, 1>),
                        reinterpret_cast<const void*>(&k_lstm_rec_tw<8, 1>), reinterpret_cast<const void*>(&k_lstm_rec_tw<2, 5>),
                        reinterpret_cast<const void*>(&k_lstm_rec_tw<4, 5>), reinterpret_cast<const void*>(&k_lstm_rec_tw<8, 5>)}) {
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)REC_LDS_CAP);
    if (e != hipSuccess && first == hipSuccess) first = e;
  }
  for (const void* f : {reinterpret_cast<const void*>(&k_lstm_rec_proj<1, 0>), reinterpret_cast<const void*>(&k_lstm_rec_proj<2, 0>),
                        reinterpret_cast<const void*>(&k_lstm_rec_proj<4, 0>), reinterpret_cast<const void*>(&k_lstm_rec_proj<8, 0>),
                        reinterpret_cast<const void*>(&k_lstm_rec_proj<1, 1>), reinterpret_cast<const void*>(&k_lstm_rec_proj<2, 1>),
                        reinterpret_cast<const void*>(&k_lstm_rec_proj<4, 1>), reinterpret_cast<const void*>(&k_lstm_rec_proj<8, 1>),
                        reinterpret_cast<const void*>(&k_lstm_rec_proj<1, 2>), reinterpret_cast<const void*>(&k_lstm_rec_proj<2, 2>),
                        reinterpret_cast<const void*>(&k_lstm_rec_proj<4, 2>), reinterpret_cast<const void*>(&k_lstm_rec_proj<8, 2>)}) {
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
    if (e != hipSuccess && first == hipSuccess) first = e;
  }
  return first;
}

void launch_lstm_rec(const RecArgs& a, int F, int rows_per_block, hipStream_t s) {
  if (F == 0) launch_f<0>(a, rows_per_block, s);
  else if (F == 1) launch_f<1>(a, rows_per_block, s);
  else launch_f<5>(a, rows_per_block, s);
}
