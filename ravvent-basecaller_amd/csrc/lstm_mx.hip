// K1m -- the BiLSTM recurrence on the 16-bit matrix pipe, SIXTEEN chunks per workgroup (replaces the tf.while_loop of LSTMCell
// steps that Encoder.call drives: /root/reference/basecaller.py:19-32,48-59; cell math SURVEY.md A.1/A.2).
//
// The packed-FMA kernels of lstm_rec.hip cost the same per chunk-row however many rows a workgroup holds (2,390 cycles per
// step for 2 rows, 8,400 for 8): they are bound by fp32 FMA issue.  Here the recurrent product of a step is ONE matrix
// product per workgroup, z^T [512 gate columns x 16 chunks] = U^T [512 x 128] . h^T [128 x 16], on v_mfma_f32_16x16x32_f16 with
// split operands (two f16 parts per value, three exact part products, f32 accumulation: DESIGN.md section 4) -- its cost does
// not depend on how many of the 16 columns are in use, so a workgroup takes 16 chunks of ONE direction:
//   * U^T stays resident as A fragments: wave w owns units 16 w .. 16 w + 15 of all four gates = 4 row tiles x 4 k-steps x
//     2 parts x 4 VGPRs = 128 VGPRs (the whole kernel, 256 KB, in the register file of the CU -- as the FMA kernels keep it);
//   * h^T goes through LDS as B fragments: [part][k-block of 8 units][chunk][8 f16], double-buffered, 1 KB per wave read;
//   * the C/D layout of the 16x16 tile gives lane (chunk n = lane % 16, q = lane / 16) the four gate sums of units
//     16 w + 4 q + 0..3 of chunk n: the cell update is local to the lane (no reduction, no gather, no tail waves), c stays in 4
//     registers, and the new h leaves as one float4 to the layer output and as 4 + 4 f16 to the LDS image;
//   * one raw s_barrier per step (LDS hand-off only: the next step's pre-projected inputs stay in flight across it).
// Scales: h' = 2^14 h (|h| <= 1), U' = s_r U with s_r the power of two that brings the largest element of gate column r into
// [2^13, 2^14); z = acc * (2^-14 / s_r) + (x . W + b).  Layer 0 with one input feature (raw samples) adds x_t W + b in the lane from
// an LDS-staged window; every other layer reads its pre-projected inputs xw [B,T,2,512] (event layer 0: k_inproj_small; layers
// >= 1: the split-f16 GEMM of gemm_f32.hip), one float4 per gate per lane, requested a step ahead.
// The reverse direction walks t = T-1..0 and, as in the reference (no mask is passed, basecaller.py:400,403), starts on the padding.
//
// CH = 8 (round 4; the per-call choice `wide_recurrence = -1` takes it for slabs that leave the chip idle): EIGHT chunks per workgroup for
// latency.  The product's 16 columns then hold both f16 PARTS of a chunk's h -- high part in column n, low part in column n + 8 -- so a
// (row tile, k-step) takes TWO MFMAs (U_hi . [h_hi | h_lo], U_lo . [h_hi | h_lo]: the three exact part products plus the low x low term,
// below 2^-22 of the product) instead of three, one LDS read instead of two; lanes n and n + 8 add their accumulators (DPP row_ror:8)
// and take two of the four units each: half the cell update per lane.  2.0 k cycles per step against 3.15 k, on twice the workgroups:
// more CU-time per chunk, which is why the streamed path keeps sixteen.  The sums associate differently: results agree with CH = 16 to
// fp32 rounding, not bit for bit.
#include "common.h"

namespace {

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define RV_MX_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

// The sixteen-chunk form with pre-projected inputs or one raw feature, as round 3 wrote it: the default path's kernels.  (The generic body
// below -- eight chunks per workgroup, five input features -- compiles to the same instruction counts for these two cases and ran 1-3 % slower
// on the same box, same inputs: 0.3945 -> 0.3990 ms for raw layer 0, 0.4035 -> 0.4155 for layer 1; the scheduling differs, not the work.)
template <int F>
__device__ __forceinline__ void rec_mx16_body(const RecArgs& a) {
  static_assert(F == 0 || F == 1, "pre-projected inputs, or one raw feature");
  extern __shared__ __align__(16) char mxsm[];
  char* hb = mxsm;                                               // [2 buffers][2 parts][16 k-blocks][16 chunks][8 f16] = 16 KB
  float* dss = reinterpret_cast<float*>(mxsm + 16384);           // [512] 2^-14 / s_r
  float* wxs = dss + RV_G;                                       // F == 1: [512] input kernel row, [512] bias
  float* xs = wxs + 2 * RV_G;                                    // F == 1: [16][T] input windows

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, q = lane >> 4;
  const int dir = blockIdx.y, b0 = blockIdx.x * RV_MX_ROWS, T = a.T;
  const int bc = min(b0 + n, a.B - 1);                           // rows beyond the slab compute on a copy of its last chunk, never stored
  const bool live = b0 + n < a.B;
  const int u0 = 16 * w + 4 * q;                                 // this lane's 4 units

  // ---- U^T -> registers (A fragments), once
  float4 ua[4][4][2];
  {
    const float4* src = reinterpret_cast<const float4*>(a.Ua[dir]) + (size_t)w * (4 * 4 * 2 * 64) + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int p = 0; p < 2; ++p) ua[g][ks][p] = src[((g * 4 + ks) * 2 + p) * 64];
  }
  {
    const float* dsg = reinterpret_cast<const float*>(a.Ua[dir] + (size_t)2 * RV_U * RV_G);
    dss[tid] = dsg[tid];
    if (F == 1) {
      wxs[tid] = a.W[dir][tid]; wxs[RV_G + tid] = a.bias[dir][tid];
      const float* xg = a.ptab ? static_cast<const float*>(a.ptab[RV_PTAB_RAW]) : a.x;     // (graph replay: the caller's address of this call)
      for (int r = 0; r < RV_MX_ROWS; ++r) {
        const int b = min(b0 + r, a.B - 1);
        const bool wm = a.mask && dir == 0 && b0 + r < a.B;     // utils.input_mask of the raw part (utils.py:26-32), once per chunk
        for (int i = tid; i < T; i += 512) {
          const float v = xg[(size_t)b * T + i];
          xs[r * T + i] = v;
          if (wm) a.mask[(size_t)b * a.mask_T + a.mask_t0 + i] = v != a.pad ? 1 : 0;
        }
      }
    }
  }
  // ---- initial state: c in registers, h as the first B image
  float c[4];
  {
    float4 h0 = make_float4(0.f, 0.f, 0.f, 0.f), c0 = h0;
    if (a.h0[dir]) { h0 = *reinterpret_cast<const float4*>(a.h0[dir] + (size_t)bc * RV_U + u0); c0 = *reinterpret_cast<const float4*>(a.c0[dir] + (size_t)bc * RV_U + u0); }
    c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w;
    const float hv[4] = {h0.x, h0.y, h0.z, h0.w};
    h4 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float sv = hv[i] * 16384.f; hi[i] = (_Float16)sv; lo[i] = (_Float16)(sv - (float)hi[i]); }
    char* dst = hb + ((2 * w + (q >> 1)) * 16 + n) * 16 + (q & 1) * 8;
    *reinterpret_cast<h4*>(dst) = hi; *reinterpret_cast<h4*>(dst + 4096) = lo;
  }
  // pre-projected inputs: the loads of step s + 2 are issued in step s (one step, ~1.3 us, is less than an HBM round trip under
  // load) into a ring of three register sets; the step loop is unrolled by three so that no set is ever COPIED -- a register
  // move from a load's destination waits for the load, which is what made a "prefetch" into a staging set synchronous
  float4 xc[4], xn[4], xnn[4];
  const float* xrow = F == 0 ? a.x + ((size_t)bc * T * 2 + dir) * RV_G + u0 : nullptr;    // + t * 1024 + g * 128
  if (F == 0) {
    const int t0 = dir ? T - 1 : 0, t1 = dir ? max(T - 2, 0) : min(1, T - 1);
#pragma unroll
    for (int g = 0; g < 4; ++g) xc[g] = *reinterpret_cast<const float4*>(xrow + (size_t)t0 * (2 * RV_G) + g * RV_U);
#pragma unroll
    for (int g = 0; g < 4; ++g) xn[g] = *reinterpret_cast<const float4*>(xrow + (size_t)t1 * (2 * RV_G) + g * RV_U);
  }
  __syncthreads();

  float hl[4] = {0.f, 0.f, 0.f, 0.f};
  int cur = 0;
  // diagnostic (RV_REC_STAMPS, tools/rec_stamps.py): cycle sums of workgroup (0, 0), wave 0 over all steps: [0] top of step -> gate sums in
  // registers, [1] -> cell update done, [2] -> LDS image and output store issued, [3] -> behind the barrier
#ifdef RV_MX_STAMPS      // diagnostic builds only (make mxvar V=1 MXFLAGS=-DRV_MX_STAMPS; RAVVENT_HIP_LIB=...libravvent_hip_m1.so): four scalar branches per step otherwise
  const bool stamp = a.dbg_ts != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
#else
  constexpr bool stamp = false;
#endif
  long long st_sum[4] = {0, 0, 0, 0};
  auto step = [&](int s, const float4 (&xu)[4], float4 (&xl)[4]) {
    const int t = dir ? T - 1 - s : s;
    long long st0 = 0;
    if (stamp) st0 = __builtin_readcyclecounter();
    if (F == 0) {                                                // the inputs of step s + 2: in flight across two barriers
      const int tn = dir ? max(t - 2, 0) : min(t + 2, T - 1);
#pragma unroll
      for (int g = 0; g < 4; ++g) xl[g] = *reinterpret_cast<const float4*>(xrow + (size_t)tn * (2 * RV_G) + g * RV_U);
    }
    f4v acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f4v{0.f, 0.f, 0.f, 0.f};
    const char* hp = hb + cur * 8192 + (q * 16 + n) * 16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const h8 bh = *reinterpret_cast<const h8*>(hp + ks * 1024), bl = *reinterpret_cast<const h8*>(hp + 4096 + ks * 1024);
      // the four gates in turns for each part product: an MFMA accumulates onto a result that is four instructions old, not onto the one
      // issued just before it (per accumulator the order of the additions is unchanged: identical results)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ua[g][ks][0]), bl, acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ua[g][ks][1]), bh, acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ua[g][ks][0]), bh, acc[g], 0, 0, 0);
    }
    // ---- gate pre-activations of this lane's 4 units, cell update (i, f, c~, o)
    float z[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 ds = *reinterpret_cast<const float4*>(&dss[g * RV_U + u0]);
      float4 xin;
      if (F == 0) xin = xu[g];
      else {
        const float xv = xs[n * T + t];
        const float4 wv = *reinterpret_cast<const float4*>(&wxs[g * RV_U + u0]), bv = *reinterpret_cast<const float4*>(&wxs[RV_G + g * RV_U + u0]);
        xin = make_float4(fmaf(xv, wv.x, bv.x), fmaf(xv, wv.y, bv.y), fmaf(xv, wv.z, bv.z), fmaf(xv, wv.w, bv.w));
      }
      z[g][0] = fmaf(acc[g][0], ds.x, xin.x); z[g][1] = fmaf(acc[g][1], ds.y, xin.y);
      z[g][2] = fmaf(acc[g][2], ds.z, xin.z); z[g][3] = fmaf(acc[g][3], ds.w, xin.w);
    }
    if (stamp) { asm volatile("" :: "v"(z[0][0]), "v"(z[3][3])); const long long tn = __builtin_readcyclecounter(); st_sum[0] += tn - st0; st0 = tn; }
    h4 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // (Round 4 tried this update on 7 transcendentals instead of 10 -- c' and h' each as ONE reciprocal of a product of (1 + 2^x) terms --
      //  parity-green and no faster: 0.394 / 0.424 ms per C3 layer either way.  The step is bound by the SIMD's issue with both of its waves
      //  in the same phase -- tools/mx_stamps.py: 1.24 k cycles of MFMA phase + 1.0 k of cell update + 0.26 k + 0.7 k at the barrier -- and a
      //  two-group form that interleaves one group's MFMAs with the other's cell update inside every wave did not overlap them either
      //  (DESIGN.md section 7).)
      const float cc = fmaf(rv_sigmoid(z[1][i]), c[i], rv_sigmoid(z[0][i]) * rv_tanh(z[2][i]));
      const float hh = rv_sigmoid(z[3][i]) * rv_tanh(cc);
      c[i] = cc; hl[i] = hh;
      const float sv = hh * 16384.f;
      hi[i] = (_Float16)sv; lo[i] = (_Float16)(sv - (float)hi[i]);
    }
    if (stamp) { asm volatile("" :: "v"(hl[0]), "v"(hl[3])); const long long tn = __builtin_readcyclecounter(); st_sum[1] += tn - st0; st0 = tn; }
    {
      char* dst = hb + (cur ^ 1) * 8192 + ((2 * w + (q >> 1)) * 16 + n) * 16 + (q & 1) * 8;
      *reinterpret_cast<h4*>(dst) = hi; *reinterpret_cast<h4*>(dst + 4096) = lo;
    }
    if (live)
      *reinterpret_cast<float4*>(a.out + ((size_t)(b0 + n) * a.out_T + a.out_t0 + t) * RV_E + dir * RV_U + u0) = make_float4(hl[0], hl[1], hl[2], hl[3]);
    cur ^= 1;
    if (stamp) { const long long tn = __builtin_readcyclecounter(); st_sum[2] += tn - st0; st0 = tn; }
    RV_MX_BARRIER();
    if (stamp) { const long long tn = __builtin_readcyclecounter(); st_sum[3] += tn - st0; }
  };
  // whole triples in the loop, the one or two steps that are left after it: a conditional step INSIDE the loop makes the compiler merge
  // two histories of pending loads / stores at the loop header, and it then drains the vector-memory counter there (`s_waitcnt vmcnt(0)`
  // once per three steps: the wave waited for the previous step's output store and for inputs it had requested one step earlier)
  int s = 0;
  for (; s + 2 < T; s += 3) {
    step(s, xc, xnn);
    step(s + 1, xn, xc);
    step(s + 2, xnn, xn);
  }
  if (s < T) {
    step(s, xc, xnn);
    if (s + 1 < T) step(s + 1, xn, xc);
  }
  if (stamp) { for (int i = 0; i < 4; ++i) a.dbg_ts[i] = st_sum[i]; a.dbg_ts[4] = T; }
  if (live) {
    *reinterpret_cast<float4*>(a.hT[dir] + (size_t)(b0 + n) * RV_U + u0) = make_float4(hl[0], hl[1], hl[2], hl[3]);
    *reinterpret_cast<float4*>(a.cT[dir] + (size_t)(b0 + n) * RV_U + u0) = make_float4(c[0], c[1], c[2], c[3]);
  }
}

__device__ __forceinline__ float ror8_add(float v) {           // v + (v of lane n ^ 8 of this 16-lane row)
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));
}

template <int F, int CH>
__device__ __forceinline__ void rec_mx_body(const RecArgs& a) {
  static_assert(F == 0 || F == 1 || F == 5, "pre-projected inputs, or layer 0 on its F input features (raw samples: 1; events: 5)");
  static_assert(CH == 16 || CH == 8, "sixteen chunks per workgroup, or eight with both parts of h in the product's columns");
  constexpr bool C8 = CH == 8;
  constexpr int NU = C8 ? 2 : 4;                                 // units per lane
  extern __shared__ __align__(16) char mxsm[];
  char* hb = mxsm;                                               // [2 buffers][2 parts][16 k-blocks][16 chunks][8 f16] = 16 KB  (CH = 8: [2][16 k-blocks][8 high | 8 low columns][8 f16])
  float* dss = reinterpret_cast<float*>(mxsm + 16384);           // [512] 2^-14 / s_r
  float* wxs = dss + RV_G;                                       // F > 0: [F][512] input kernel rows, [512] bias
  float* xs = wxs + (F + 1) * RV_G;                              // F > 0: [CH][T][F] input windows

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, q = lane >> 4;
  const int nn = C8 ? n & 7 : n, hs = C8 ? n >> 3 : 0;           // chunk of the group; CH = 8: which half of the four units (and: high / low column)
  const int dir = blockIdx.y, b0 = blockIdx.x * CH, T = a.T;
  const int bc = min(b0 + nn, a.B - 1);                          // rows beyond the slab compute on a copy of its last chunk, never stored
  const bool live = b0 + nn < a.B;
  const int u0 = 16 * w + 4 * q + 2 * hs;                        // this lane's NU units

  // ---- U^T -> registers (A fragments), once
  float4 ua[4][4][2];
  {
    const float4* src = reinterpret_cast<const float4*>(a.Ua[dir]) + (size_t)w * (4 * 4 * 2 * 64) + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int p = 0; p < 2; ++p) ua[g][ks][p] = src[((g * 4 + ks) * 2 + p) * 64];
  }
  {
    const float* dsg = reinterpret_cast<const float*>(a.Ua[dir] + (size_t)2 * RV_U * RV_G);
    dss[tid] = dsg[tid];
    if (F > 0) {
#pragma unroll
      for (int f = 0; f < F; ++f) wxs[f * RV_G + tid] = a.W[dir][f * RV_G + tid];
      wxs[F * RV_G + tid] = a.bias[dir][tid];
      const float* xg = a.ptab ? static_cast<const float*>(a.ptab[F == 1 ? RV_PTAB_RAW : RV_PTAB_EVENT]) : a.x;     // (graph replay: the caller's address of this call)
      for (int r = 0; r < CH; ++r) {
        const int b = min(b0 + r, a.B - 1);
        const bool wm = a.mask && dir == 0 && b0 + r < a.B;     // utils.input_mask of this encoder's part (utils.py:26-32: all features != pad), once per chunk
        for (int i = tid; i < T; i += 512) {
          bool real = true;
#pragma unroll
          for (int f = 0; f < F; ++f) { const float v = xg[((size_t)b * T + i) * F + f]; xs[(r * T + i) * F + f] = v; real = real && v != a.pad; }
          if (wm) a.mask[(size_t)b * a.mask_T + a.mask_t0 + i] = real ? 1 : 0;
        }
      }
    }
  }
  // ---- initial state: c in registers, h as the first B image
  float c[NU];
  // the h image a lane writes: its NU units of chunk nn as f16 parts.  CH = 16: [part][k-block][chunk][8]; CH = 8: [k-block][column][8] with the
  // high part in column nn and the low part in column nn + 8
  auto put_h = [&](int buf, const float (&hv)[NU]) {
    _Float16 hi[NU], lo[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) { const float sv = hv[i] * 16384.f; hi[i] = (_Float16)sv; lo[i] = (_Float16)(sv - (float)hi[i]); }
    if constexpr (C8) {
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      char* dst = hb + buf * 4096 + ((2 * w + (q >> 1)) * 16 + nn) * 16 + (q & 1) * 8 + hs * 4;
      *reinterpret_cast<h2*>(dst) = h2{hi[0], hi[1]}; *reinterpret_cast<h2*>(dst + 8 * 16) = h2{lo[0], lo[1]};
    } else {
      char* dst = hb + buf * 8192 + ((2 * w + (q >> 1)) * 16 + n) * 16 + (q & 1) * 8;
      *reinterpret_cast<h4*>(dst) = h4{hi[0], hi[1], hi[2], hi[3]}; *reinterpret_cast<h4*>(dst + 4096) = h4{lo[0], lo[1], lo[2], lo[3]};
    }
  };
  {
    float hv[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) { hv[i] = 0.f; c[i] = 0.f; }
    if (a.h0[dir]) {
#pragma unroll
      for (int i = 0; i < NU; ++i) { hv[i] = a.h0[dir][(size_t)bc * RV_U + u0 + i]; c[i] = a.c0[dir][(size_t)bc * RV_U + u0 + i]; }
    }
    put_h(0, hv);
  }
  // pre-projected inputs: the loads of step s + 2 are issued in step s (one step, ~1.3 us, is less than an HBM round trip under
  // load) into a ring of three register sets; the step loop is unrolled by three so that no set is ever COPIED -- a register
  // move from a load's destination waits for the load, which is what made a "prefetch" into a staging set synchronous
  struct XV { float v[NU]; };                                    // NU consecutive floats (one 16- / 8-byte load)
  auto ldx = [&](const float* p_) { XV r; if constexpr (C8) { const float2 t_ = *reinterpret_cast<const float2*>(p_); r.v[0] = t_.x; r.v[1] = t_.y; }
                                    else { const float4 t_ = *reinterpret_cast<const float4*>(p_); r.v[0] = t_.x; r.v[1] = t_.y; r.v[2] = t_.z; r.v[3] = t_.w; } return r; };
  XV xc[4], xn[4], xnn[4];
  const float* xrow = F == 0 ? a.x + ((size_t)bc * T * 2 + dir) * RV_G + u0 : nullptr;    // + t * 1024 + g * 128
  if (F == 0) {
    const int t0 = dir ? T - 1 : 0, t1 = dir ? max(T - 2, 0) : min(1, T - 1);
#pragma unroll
    for (int g = 0; g < 4; ++g) xc[g] = ldx(xrow + (size_t)t0 * (2 * RV_G) + g * RV_U);
#pragma unroll
    for (int g = 0; g < 4; ++g) xn[g] = ldx(xrow + (size_t)t1 * (2 * RV_G) + g * RV_U);
  }
  __syncthreads();

  float hl[NU];
#pragma unroll
  for (int i = 0; i < NU; ++i) hl[i] = 0.f;
  int cur = 0;
  // diagnostic (RV_REC_STAMPS, tools/rec_stamps.py): cycle sums of workgroup (0, 0), wave 0 over all steps: [0] top of step -> gate sums in
  // registers, [1] -> cell update done, [2] -> LDS image and output store issued, [3] -> behind the barrier
#ifdef RV_MX_STAMPS      // diagnostic builds only (make mxvar V=1 MXFLAGS=-DRV_MX_STAMPS; RAVVENT_HIP_LIB=...libravvent_hip_m1.so): four scalar branches per step otherwise
  const bool stamp = a.dbg_ts != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
#else
  constexpr bool stamp = false;
#endif
  long long st_sum[4] = {0, 0, 0, 0};
  auto step = [&](int s, const XV (&xu)[4], XV (&xl)[4]) {
    const int t = dir ? T - 1 - s : s;
    long long st0 = 0;
    if (stamp) st0 = __builtin_readcyclecounter();
    if (F == 0) {                                                // the inputs of step s + 2: in flight across two barriers
      const int tn = dir ? max(t - 2, 0) : min(t + 2, T - 1);
#pragma unroll
      for (int g = 0; g < 4; ++g) xl[g] = ldx(xrow + (size_t)tn * (2 * RV_G) + g * RV_U);
    }
    f4v acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f4v{0.f, 0.f, 0.f, 0.f};
    if constexpr (C8) {
      const char* hp = hb + cur * 4096 + (q * 16 + n) * 16;      // column n: chunk n % 8, high (n < 8) or low part
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const h8 bb = *reinterpret_cast<const h8*>(hp + ks * 1024);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ua[g][ks][1]), bb, acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ua[g][ks][0]), bb, acc[g], 0, 0, 0);
      }
    } else {
    const char* hp = hb + cur * 8192 + (q * 16 + n) * 16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const h8 bh = *reinterpret_cast<const h8*>(hp + ks * 1024), bl = *reinterpret_cast<const h8*>(hp + 4096 + ks * 1024);
      // the four gates in turns for each part product: an MFMA accumulates onto a result that is four instructions old, not onto the one
      // issued just before it (per accumulator the order of the additions is unchanged: identical results)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ua[g][ks][0]), bl, acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ua[g][ks][1]), bh, acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ua[g][ks][0]), bh, acc[g], 0, 0, 0);
    }
    }
    // ---- gate pre-activations of this lane's NU units, cell update (i, f, c~, o)
    float z[4][NU];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float sacc[NU];
      if constexpr (C8) {                                        // columns n and n + 8 = [U . h_hi] and [U . h_lo] of one chunk: add; this lane keeps units 2 hs, + 1
        const float s0 = ror8_add(acc[g][0]), s1 = ror8_add(acc[g][1]), s2 = ror8_add(acc[g][2]), s3 = ror8_add(acc[g][3]);
        sacc[0] = hs ? s2 : s0; sacc[1] = hs ? s3 : s1;
      } else {
#pragma unroll
        for (int i = 0; i < NU; ++i) sacc[i] = acc[g][i];
      }
      const XV ds = ldx(&dss[g * RV_U + u0]);
      XV xin;
      if (F == 0) xin = xu[g];
      else {                                                     // b + sum_f x_f w_f, features in order (the same additions as k_inproj_small: identical bits)
        xin = ldx(&wxs[F * RV_G + g * RV_U + u0]);
#pragma unroll
        for (int f = 0; f < F; ++f) {
          const float xv = xs[(nn * T + t) * F + f];
          const XV wv = ldx(&wxs[f * RV_G + g * RV_U + u0]);
#pragma unroll
          for (int i = 0; i < NU; ++i) xin.v[i] = fmaf(xv, wv.v[i], xin.v[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) z[g][i] = fmaf(sacc[i], ds.v[i], xin.v[i]);
    }
    if (stamp) { asm volatile("" :: "v"(z[0][0]), "v"(z[3][NU - 1])); const long long tn = __builtin_readcyclecounter(); st_sum[0] += tn - st0; st0 = tn; }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      // (Round 4 tried this update on 7 transcendentals instead of 10 -- c' and h' each as ONE reciprocal of a product of (1 + 2^x) terms --
      //  parity-green and no faster: 0.394 / 0.424 ms per C3 layer either way.  The step is bound by the SIMD's issue with both of its waves
      //  in the same phase -- tools/mx_stamps.py: 1.24 k cycles of MFMA phase + 1.0 k of cell update + 0.26 k + 0.7 k at the barrier -- and a
      //  two-group form that interleaves one group's MFMAs with the other's cell update inside every wave did not overlap them either
      //  (DESIGN.md section 7).)
      const float cc = fmaf(rv_sigmoid(z[1][i]), c[i], rv_sigmoid(z[0][i]) * rv_tanh(z[2][i]));
      const float hh = rv_sigmoid(z[3][i]) * rv_tanh(cc);
      c[i] = cc; hl[i] = hh;
    }
    if (stamp) { asm volatile("" :: "v"(hl[0]), "v"(hl[NU - 1])); const long long tn = __builtin_readcyclecounter(); st_sum[1] += tn - st0; st0 = tn; }
    put_h(cur ^ 1, hl);
    if (live) {
      float* op = a.out + ((size_t)(b0 + nn) * a.out_T + a.out_t0 + t) * RV_E + dir * RV_U + u0;
      if constexpr (C8) *reinterpret_cast<float2*>(op) = make_float2(hl[0], hl[1]);
      else *reinterpret_cast<float4*>(op) = make_float4(hl[0], hl[1], hl[2], hl[3]);
    }
    cur ^= 1;
    if (stamp) { const long long tn = __builtin_readcyclecounter(); st_sum[2] += tn - st0; st0 = tn; }
    RV_MX_BARRIER();
    if (stamp) { const long long tn = __builtin_readcyclecounter(); st_sum[3] += tn - st0; }
  };
  // whole triples in the loop, the one or two steps that are left after it: a conditional step INSIDE the loop makes the compiler merge
  // two histories of pending loads / stores at the loop header, and it then drains the vector-memory counter there (`s_waitcnt vmcnt(0)`
  // once per three steps: the wave waited for the previous step's output store and for inputs it had requested one step earlier)
  int s = 0;
  for (; s + 2 < T; s += 3) {
    step(s, xc, xnn);
    step(s + 1, xn, xc);
    step(s + 2, xnn, xn);
  }
  if (s < T) {
    step(s, xc, xnn);
    if (s + 1 < T) step(s + 1, xn, xc);
  }
  if (stamp) { for (int i = 0; i < 4; ++i) a.dbg_ts[i] = st_sum[i]; a.dbg_ts[4] = T; }
  if (live) {
#pragma unroll
    for (int i = 0; i < NU; ++i) { a.hT[dir][(size_t)(b0 + nn) * RV_U + u0 + i] = hl[i]; a.cT[dir][(size_t)(b0 + nn) * RV_U + u0 + i] = c[i]; }
  }
}

template <int F, int CH>
__global__ __launch_bounds__(512) void k_lstm_rec_mx(RecArgs a) {
  if constexpr (CH == 16 && F != 5) rec_mx16_body<F>(a);
  else rec_mx_body<F, CH>(a);
}

// x . W + b for a layer-0 encoder with a handful of input features (the event encoder: F = 5), both directions:
// xw [B*T, 2, 512].  One thread = 4 gate columns of one row; HBM-bound on its output (4 KB per chunk-timestep).
template <int F>
__global__ __launch_bounds__(256) void k_inproj_small(const float* __restrict__ x_arg, int rows, const float* __restrict__ W0, const float* __restrict__ b0,
                                                       const float* __restrict__ W1, const float* __restrict__ b1, float* __restrict__ xw,
                                                       uint8_t* __restrict__ mask, int T, int mask_T, int mask_t0, float pad, const void* const* xtab) {
  const float* __restrict__ x = xtab ? static_cast<const float*>(xtab[F == 1 ? RV_PTAB_RAW : RV_PTAB_EVENT]) : x_arg;
  const int c4 = threadIdx.x & 127, dir = threadIdx.x >> 7;
  const float* W = dir ? W1 : W0;
  const float* b = dir ? b1 : b0;
  float4 wr[F];
#pragma unroll
  for (int f = 0; f < F; ++f) wr[f] = *reinterpret_cast<const float4*>(W + (size_t)f * RV_G + 4 * c4);
  const float4 bv = *reinterpret_cast<const float4*>(b + 4 * c4);
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    float4 acc = bv;
    bool real = true;                                            // utils.input_mask (utils.py:26-32): all features != pad
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const float xv = x[(size_t)r * F + f];
      real = real && xv != pad;
      acc.x = fmaf(xv, wr[f].x, acc.x); acc.y = fmaf(xv, wr[f].y, acc.y); acc.z = fmaf(xv, wr[f].z, acc.z); acc.w = fmaf(xv, wr[f].w, acc.w);
    }
    *reinterpret_cast<float4*>(xw + ((size_t)r * 2 + dir) * RV_G + 4 * c4) = acc;
    if (mask && threadIdx.x == 0) mask[(size_t)(r / T) * mask_T + mask_t0 + r % T] = real ? 1 : 0;
  }
}

constexpr size_t mx_lds_bytes(int F, int T) { return 16384 + sizeof(float) * ((size_t)RV_G + (F > 0 ? (size_t)(F + 1) * RV_G + (size_t)RV_MX_ROWS * T * F : 2 * RV_G)); }

}  // namespace

bool lstm_rec_mx_window_fits(int T, int F) { return mx_lds_bytes(F, T) <= 160 * 1024; }

hipError_t configure_mx_kernels() {
  hipError_t first = hipSuccess;
  for (const void* f : {reinterpret_cast<const void*>(&k_lstm_rec_mx<0, 16>), reinterpret_cast<const void*>(&k_lstm_rec_mx<1, 16>),
                        reinterpret_cast<const void*>(&k_lstm_rec_mx<5, 16>), reinterpret_cast<const void*>(&k_lstm_rec_mx<0, 8>),
                        reinterpret_cast<const void*>(&k_lstm_rec_mx<1, 8>), reinterpret_cast<const void*>(&k_lstm_rec_mx<5, 8>)}) {
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess && first == hipSuccess) first = e;
  }
  return first;
}

void launch_lstm_rec_mx(const RecArgs& a, int F, hipStream_t s, bool rows8) {
  if (rows8) {                                                   // eight chunks per workgroup: the latency form (see the head of this file)
    dim3 grid((a.B + 7) / 8, 2);
    if (F == 1) hipLaunchKernelGGL((k_lstm_rec_mx<1, 8>), grid, dim3(512), mx_lds_bytes(1, a.T), s, a);
    else if (F == 5) hipLaunchKernelGGL((k_lstm_rec_mx<5, 8>), grid, dim3(512), mx_lds_bytes(5, a.T), s, a);
    else hipLaunchKernelGGL((k_lstm_rec_mx<0, 8>), grid, dim3(512), mx_lds_bytes(0, a.T), s, a);
    return;
  }
  dim3 grid((a.B + RV_MX_ROWS - 1) / RV_MX_ROWS, 2);
  if (F == 1) hipLaunchKernelGGL((k_lstm_rec_mx<1, 16>), grid, dim3(512), mx_lds_bytes(1, a.T), s, a);
  else if (F == 5) hipLaunchKernelGGL((k_lstm_rec_mx<5, 16>), grid, dim3(512), mx_lds_bytes(5, a.T), s, a);
  else hipLaunchKernelGGL((k_lstm_rec_mx<0, 16>), grid, dim3(512), mx_lds_bytes(0, a.T), s, a);
}

void launch_inproj_small(const float* x, int rows, int F, const float* W0, const float* b0, const float* W1, const float* b1, float* xw,
                         uint8_t* mask, int T, int mask_T, int mask_t0, float pad, hipStream_t s, const void* const* xtab) {
  const int grid = rows < 4096 ? rows : 4096;
  if (F == 5) hipLaunchKernelGGL((k_inproj_small<5>), dim3(grid), dim3(256), 0, s, x, rows, W0, b0, W1, b1, xw, mask, T, mask_T, mask_t0, pad, xtab);
  else hipLaunchKernelGGL((k_inproj_small<1>), dim3(grid), dim3(256), 0, s, x, rows, W0, b0, W1, b1, xw, mask, T, mask_T, mask_t0, pad, xtab);
}
