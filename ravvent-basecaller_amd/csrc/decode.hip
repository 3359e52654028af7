// Decoder-side kernels: one AttentionWrapper step + output layer + beam expansion per launch
// pair, replacing the tf.while_loop body of tfa.seq2seq.BeamSearchDecoder / BasicDecoder that
// /root/reference/basecaller.py:306-313 and :322-329 drive (semantics: SURVEY.md A.3-A.6).
//
//   k_dec_gates   LSTM cell non-linearity on the pre-activations the D1 GEMM produced.
//   k_dec_attend  ONE workgroup per chunk handles all W beams of that chunk, so keys/values are
//                 read once per chunk instead of once per beam (the reference tiles them W times,
//                 tile_batch at basecaller.py:300-301): score (Luong q.k / Bahdanau v.tanh(k+Wq q)),
//                 -inf masking, softmax, context, attention layer, fc, log-softmax, finished-beam
//                 masking, top-W over W*V, and the parent-gather of every state tensor.
//   k_dec_finalize gather_tree for beam 0 + slicing of predicted_ids[:,:,0] / scores[:,:,0].
// Loop control: the reference stops when ALL rows are finished; here each chunk-workgroup counts
// itself into nfin[step] once all its beams are finished, and every kernel of step s exits early
// when nfin[s-1] == B (the launch sequence is fixed, so it can live in a hipGraph).
#include "common.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <type_traits>

namespace {

__global__ void k_input_mask(const float* raw, const float* ev, int B, int T_r, int T_e, float pad,
                             uint8_t* mask) {
  const int Tm = T_r + T_e;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * Tm) return;
  const int b = idx / Tm, t = idx % Tm;
  bool ok;
  if (t < T_r) {
    ok = raw[(size_t)b * T_r + t] != pad;            // all(x != pad) over the 1 raw feature
  } else {
    const float* e = ev + ((size_t)b * T_e + (t - T_r)) * 5;
    ok = e[0] != pad && e[1] != pad && e[2] != pad && e[3] != pad && e[4] != pad;
  }
  mask[idx] = ok ? 1 : 0;
}

__global__ void k_dec_init(DecState d) {
  const int N = d.B * d.W;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < N) {
    d.tok[idx] = d.start_token;
    d.log_probs[idx] = (idx % d.W) == 0 ? 0.f : -INFINITY;   // one_hot(0, W, on=0, off=-inf)
    d.finished[idx] = 0;
    d.lengths[idx] = 0;
  }
  if (idx < d.L) d.nfin[idx] = 0;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// D1 -- decoder LSTM cell, fused: pre-activations on v_mfma_f32_16x16x4_f32 + gate math.
// z = [attention | h] . [W_dec[V:] ; U_dec] + W_dec[token] + b  (cell input = concat(one_hot, attention),
// SURVEY.md A.4; the one-hot product is a row gather).
// Workgroup = 64 beam rows x 16 units x 4 gates, K = 256 in one shot: the 64 KB activation panel and
// the 64 KB weight panel (pre-transposed to [col][k] at load time, so both are coalesced 1 KB row
// copies) go to LDS with every load in flight -- the epilogue's token / bias / cell-state operands are
// fetched first so their round trips hide under the panels -- ONE barrier, then each of the 8 waves
// runs 2 x 64 MFMAs (16 rows x one gate's 16 units per tile, operands read as float4 along k) and the
// four gate tiles meet in LDS for the cell update.  (Measured alternatives: fragments straight from
// global to registers 14 us -- 64 cache lines per load instruction; 32-row tiles 12 us -- two
// rounds of workgroups; this form 8-10 us.)
constexpr int CELL_LD = 260;                       // padded row length of the LDS panels (floats)
constexpr int CELL_ROWS = 64;
constexpr int CELL_LDS_FLOATS = (CELL_ROWS + 64) * CELL_LD + 4 * CELL_ROWS * 17;
__global__ __launch_bounds__(512) void k_dec_cell(DecState d, int layer, const float* __restrict__ WcatT,
                                                  const float* __restrict__ Wtok, const float* __restrict__ bias,
                                                  int step) {
  if (step > 0 && d.nfin[step - 1] >= d.B) return;
  const float* xh_l = d.xh + layer * d.ls_xh;
  const float* c_l = d.c + layer * d.ls_c;
  float* cn_l = d.c_new + layer * d.ls_c;
  float* hn_l = d.h_new + layer * d.ls_c;
  float* xh_up = layer + 1 < d.depth ? d.xh + (layer + 1) * d.ls_xh : nullptr;   // next cell's input half
  extern __shared__ __align__(16) float csm[];
  float* As = csm;                                  // [64 rows][260]
  float* Bs = csm + CELL_ROWS * CELL_LD;            // [64 cols = 4 gates x 16 units][260]
  float* zs = csm + (CELL_ROWS + 64) * CELL_LD;     // [4 gates][64 rows][17]
  const int N = d.B * d.W;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r0 = blockIdx.y * CELL_ROWS, u0 = blockIdx.x * 16;
  // epilogue operands first (their round trips hide under the panel loads): thread = 2 x (row, unit)
  int en[2]; float ec[2], eb[4], ew[2][4];
  const int eun = tid & 15, ecol = u0 + eun;
#pragma unroll
  for (int gg = 0; gg < 4; ++gg) eb[gg] = bias[gg * RV_U + ecol];
  int etok[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    en[p] = min(r0 + (tid >> 4) + 32 * p, N - 1);
    etok[p] = Wtok ? d.tok[en[p]] : 0;
    ec[p] = c_l[(size_t)en[p] * RV_U + ecol];
  }
  float4 ra[8], rb[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int idx = tid + 512 * p, row = idx >> 6, c4 = idx & 63;        // 64 float4 per 1 KB row
    const int n = min(r0 + row, N - 1);
    ra[p] = *reinterpret_cast<const float4*>(xh_l + (size_t)n * RV_E + 4 * c4);
    const int col = (row >> 4) * RV_U + u0 + (row & 15);                 // panel row = gate*16 + unit
    rb[p] = *reinterpret_cast<const float4*>(WcatT + (size_t)col * RV_E + 4 * c4);
  }
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int idx = tid + 512 * p, row = idx >> 6, c4 = idx & 63;
    *reinterpret_cast<float4*>(As + row * CELL_LD + 4 * c4) = ra[p];
    *reinterpret_cast<float4*>(Bs + row * CELL_LD + 4 * c4) = rb[p];
  }
#pragma unroll
  for (int p = 0; p < 2; ++p)       // one-hot row gather; lands while the MFMAs run
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) ew[p][gg] = Wtok ? Wtok[(size_t)etok[p] * RV_G + gg * RV_U + ecol] : 0.f;
  __syncthreads();
  const int rq = wv & 3, gh = wv >> 2;            // wave = 16-row quarter x gate pair
  const int li = lane & 15, kq = lane >> 4;       // lane group kq supplies k in [64kq, 64kq+64)
  const float4* ap = reinterpret_cast<const float4*>(As + (16 * rq + li) * CELL_LD + 64 * kq);
#pragma unroll
  for (int gi = 0; gi < 2; ++gi) {
    const int g = 2 * gh + gi;
    const float4* bp = reinterpret_cast<const float4*>(Bs + (16 * g + li) * CELL_LD + 64 * kq);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float4 av = ap[i], bv = bp[i];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc1, 0, 0, 0);
    }
    // C/D map of the 16x16 tile: col = lane & 15, row = 4*(lane >> 4) + reg
#pragma unroll
    for (int r = 0; r < 4; ++r) zs[(g * CELL_ROWS + 16 * rq + 4 * kq + r) * 17 + li] = acc0[r] + acc1[r];
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int row = (tid >> 4) + 32 * p;
    if (r0 + row < N) {
      const float zi = zs[(0 * CELL_ROWS + row) * 17 + eun] + ew[p][0] + eb[0];
      const float zf = zs[(1 * CELL_ROWS + row) * 17 + eun] + ew[p][1] + eb[1];
      const float zg = zs[(2 * CELL_ROWS + row) * 17 + eun] + ew[p][2] + eb[2];
      const float zo = zs[(3 * CELL_ROWS + row) * 17 + eun] + ew[p][3] + eb[3];
      const float c2 = fmaf(rv_sigmoid(zf), ec[p], rv_sigmoid(zi) * rv_tanh(zg));
      const float hh = rv_sigmoid(zo) * rv_tanh(c2);
      cn_l[(size_t)en[p] * RV_U + ecol] = c2;
      hn_l[(size_t)en[p] * RV_U + ecol] = hh;
      if (xh_up) xh_up[(size_t)en[p] * RV_E + ecol] = hh;
    }
  }
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
// 8 floats -> two f16 fragments of v * scale: hi = f16(s), lo = f16(s - hi) (the residual is exact in f32)
__device__ __forceinline__ void split_f16x8(const float* v, float scale, float4& hi, float4& lo) {
  h8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float sv = v[j] * scale;
    a[j] = (_Float16)sv;
    b[j] = (_Float16)(sv - (float)a[j]);
  }
  hi = __builtin_bit_cast(float4, a); lo = __builtin_bit_cast(float4, b);
}

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row; every lane ends with the total
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp<0x141>(v);   // row_half_mirror
  v += dpp<0x140>(v);   // row_mirror
  return v;
}
// value of lane n (0..7) of the caller's 16-lane row, in every lane of that row (DPP row_newbcast)
__device__ __forceinline__ float row_bcast(float v, int n) {
  switch (n) {
    case 0: return dpp<0x150>(v); case 1: return dpp<0x151>(v); case 2: return dpp<0x152>(v);
    case 3: return dpp<0x153>(v); case 4: return dpp<0x154>(v); case 5: return dpp<0x155>(v);
    case 6: return dpp<0x156>(v); default: return dpp<0x157>(v);
  }
}
// same, source lane n in 0..15
__device__ __forceinline__ float row_bcast16(float v, int n) {
  switch (n) {
    case 0: return dpp<0x150>(v); case 1: return dpp<0x151>(v); case 2: return dpp<0x152>(v); case 3: return dpp<0x153>(v);
    case 4: return dpp<0x154>(v); case 5: return dpp<0x155>(v); case 6: return dpp<0x156>(v); case 7: return dpp<0x157>(v);
    case 8: return dpp<0x158>(v); case 9: return dpp<0x159>(v); case 10: return dpp<0x15A>(v); case 11: return dpp<0x15B>(v);
    case 12: return dpp<0x15C>(v); case 13: return dpp<0x15D>(v); case 14: return dpp<0x15E>(v); default: return dpp<0x15F>(v);
  }
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// wave-uniform max via DPP inside the four 16-lane rows + readlane across them (no LDS round trips)
__device__ __forceinline__ float wave_max_fast(float v) {
  v = fmaxf(v, dpp<0xB1>(v));
  v = fmaxf(v, dpp<0x4E>(v));
  v = fmaxf(v, dpp<0x141>(v));
  v = fmaxf(v, dpp<0x140>(v));
  const int vi = __float_as_int(v);
  const float a = __int_as_float(__builtin_amdgcn_readlane(vi, 0)), b = __int_as_float(__builtin_amdgcn_readlane(vi, 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(vi, 32)), e = __int_as_float(__builtin_amdgcn_readlane(vi, 48));
  return fmaxf(fmaxf(a, b), fmaxf(c, e));
}

// wave-uniform sum, fixed order: DPP inside the 16-lane rows, then rows 0..3 left to right
__device__ __forceinline__ float wave_sum_fast(float v) {
  v = row16_sum(v);
  const int vi = __float_as_int(v);
  const float a = __int_as_float(__builtin_amdgcn_readlane(vi, 0)), b = __int_as_float(__builtin_amdgcn_readlane(vi, 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(vi, 32)), e = __int_as_float(__builtin_amdgcn_readlane(vi, 48));
  return ((a + b) + c) + e;
}

constexpr int WB = RV_MAX_BEAM;
constexpr int ATT_THREADS = 512;
#ifndef RV_STAMP_WAVE
#define RV_STAMP_WAVE 1      // which wave stamps its cell product (diagnostic builds may pick another: make decvar DECFLAGS=-DRV_STAMP_WAVE=4)
#endif
#define RV_STAMP_W1(d, step, i) do { if ((d).dbg_ts && blockIdx.x == 0 && threadIdx.x == 64 * RV_STAMP_WAVE && (step) == 3) (d).dbg_ts[i] = __builtin_readcyclecounter(); } while (0)
#define RV_STAMP(d, step, i) do { if ((d).dbg_ts && blockIdx.x == 0 && threadIdx.x == 0 && (step) == 3) (d).dbg_ts[i] = __builtin_readcyclecounter(); } while (0)
constexpr float LOG2E = 1.4426950408889634f;
// Matrix-pipe forms of the persistent decode: the A operand of every product (query, alignments, [ctx' | h]) has the beams as rows,
// at most 8 of an MFMA tile's 16.  Both f16 PARTS of a beam's value therefore ride in ONE tile -- beam w: high part in row
// 4 (w % 4) + 2 (w / 4), low part in the row after it -- so a product is TWO MFMAs per k-step (the tile against the high and against the
// low part of B) instead of three, and one LDS read per k-step instead of two: rows r, r + 1 of `. B_hi` and row r of `. B_lo` are the
// three exact part products (row r + 1 of `. B_lo` is the low x low term, below 2^-22 of the product: it is simply added too).
// C/D map of the 16x16 tile: lane (column l % 16, g = l / 16) holds rows 4 g + i, i = 0..3: registers 0 + 1 = beam g, 2 + 3 = beam g + 4.
__host__ __device__ constexpr int mx_row(int w) { return 4 * (w & 3) + 2 * (w >> 2); }

// Static LDS shared by both attend kernels: output layer, new cell states, beam bookkeeping.
struct AttShared {
  float wfc[RV_U * RV_MAX_VOCAB + RV_MAX_VOCAB];   // W_fc [128][V] then b_fc [V]
  float cnew[WB * RV_U];
  float lprob[WB];
  int fin[WB], len[WB], parent[WB];
};

// Dynamic LDS carve-up (floats), shared with the host-side size computation.
struct AttLds {
  int q, pq, sc, al, part, hcT, att, lg, fold, total;
  __host__ __device__ AttLds(int W, int TmP, bool flash, int NT = 512) {
    const int NW = NT / 64;                     // waves per workgroup
    int o = 0;
    q = o; o += W * RV_U;
    pq = o; o += flash ? W * RV_E : W * RV_U;   // Bahdanau processed query [W][128] / flash q' [W][256]
    sc = o; o += flash ? (NT / 16) * WB * 2 : W * TmP; // two-pass: scores [W][TmP]; flash: per-stream (max, sum) [NT/16][WB][2]
    al = o; o += flash ? 2 * WB : TmP * WB;     // two-pass: alignments [TmP][WB]; flash: merged (max, 1/sum)
    part = o; o += NW * W * RV_E;               // partial sums [NW][W][256] (also [2 NW][W][128])
    hcT = o; o += (RV_U + RV_E) * WB;           // [h ; context] k-major, beam-minor
    att = o; o += W * RV_U;
    lg = o; o += WB * RV_MAX_VOCAB;
    fold = o; o += flash ? NW * W * 4 * (NT == 512 ? 32 : 16) * 4 : 0;  // flash: wave-private fold slabs [NW][W*4][32|16] float4
    total = o;
  }
};

// ---- phase A: one round of small loads -> LDS: query = cell output h of every beam (also rows
// 0..127 of hcT), the new cell states, the beam bookkeeping and the output layer.
template <int W, int NT>
__device__ __forceinline__ void att_prologue(const DecState& d, AttShared& S, float* q, float* hcT, size_t row0, int tid) {
  constexpr int ATT_THREADS = NT;
  const int V = d.V;
  const float* hn_top = d.h_new + (d.depth - 1) * d.ls_c;
  const float* cn_top = d.c_new + (d.depth - 1) * d.ls_c;
  for (int i = tid; i < W * RV_U; i += ATT_THREADS) {
    const float v = hn_top[row0 * RV_U + i];
    q[i] = v;
    hcT[(i & 127) * WB + (i >> 7)] = v;
    S.cnew[i] = cn_top[row0 * RV_U + i];
  }
  for (int i = tid; i < RV_U * V; i += ATT_THREADS) S.wfc[i] = d.W_fc[i];
  if (tid < V) S.wfc[RV_U * V + tid] = d.b_fc[tid];
  if (tid < W) {
    S.fin[tid] = d.finished[row0 + tid];
    S.lprob[tid] = d.log_probs[row0 + tid];
    S.len[tid] = d.lengths[row0 + tid];
  }
}

// ---- phases E-H: attention layer, output layer, sampler / beam step, parent-gather of the state.
// Expects hcT = [h ; context] complete and a barrier behind it.
template <int W, int NT>
__device__ __forceinline__ void att_tail(const DecState& d, AttShared& S, const float* q, const float* hcT, float* part,
                                         float* att, float* lg, int b, int step, int tid, bool all_fin = false) {
  constexpr int ATT_THREADS = NT;
  constexpr int NKG = NT / 32;              // K-groups of the attention layer: 16 x 24 rows (NT=512) / 8 x 48 (NT=256)
  constexpr int KPG = 384 / NKG;
  const int V = d.V;
  const size_t row0 = (size_t)b * W;
  if (!all_fin) {     // a chunk whose beams are all finished needs no logits (see the caller)
  // E: attention = [h ; context] . W_att   (Dense, no bias, no activation)
  //    thread = (4 output columns, 1 of NKG K-groups of KPG rows); 24 float4 weight loads in flight
  {
    const int d4 = tid & 31, kg = tid >> 5;
    f2 acc[W][2];
#pragma unroll
    for (int w = 0; w < W; ++w) { acc[w][0] = f2{0.f, 0.f}; acc[w][1] = f2{0.f, 0.f}; }
#pragma unroll
    for (int k0 = 0; k0 < KPG; k0 += 24) {
      float4 wv[24];
      const float* wa = d.W_att + (size_t)(KPG * kg + k0) * RV_U + 4 * d4;
#pragma unroll
      for (int u = 0; u < 24; ++u) wv[u] = *reinterpret_cast<const float4*>(wa + (size_t)u * RV_U);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 24; ++u) {
        float hv[WB];
        *reinterpret_cast<float4*>(hv) = *reinterpret_cast<const float4*>(&hcT[(KPG * kg + k0 + u) * WB]);
        if (W > 4) *reinterpret_cast<float4*>(hv + 4) = *reinterpret_cast<const float4*>(&hcT[(KPG * kg + k0 + u) * WB + 4]);
#pragma unroll
        for (int w = 0; w < W; ++w) {
          acc[w][0] = __builtin_elementwise_fma(f2{hv[w], hv[w]}, f2{wv[u].x, wv[u].y}, acc[w][0]);
          acc[w][1] = __builtin_elementwise_fma(f2{hv[w], hv[w]}, f2{wv[u].z, wv[u].w}, acc[w][1]);
        }
      }
    }
#pragma unroll
    for (int w = 0; w < W; ++w)
      *reinterpret_cast<float4*>(&part[(kg * W + w) * RV_U + 4 * d4]) =
          make_float4(acc[w][0].x, acc[w][0].y, acc[w][1].x, acc[w][1].y);
  }
  __syncthreads();
  for (int i = tid; i < W * RV_U; i += ATT_THREADS) {      // fixed-order reduction of the partials
    const int w = i >> 7, col = i & 127;
    float s0 = 0.f;
#pragma unroll
    for (int g = 0; g < NKG; ++g) s0 += part[(g * W + w) * RV_U + col];
    att[i] = s0;
  }
  __syncthreads();
  RV_STAMP(d, step, 5);
  if (d.dbg_stop == 5) return;

  // F: logits = attention . W_fc + b_fc ; one 16-lane row per (beam, token)
  {
    const int sub = tid & 15, o = tid >> 4;         // 32 outputs per pass
    for (int ob = o; ob < W * V; ob += ATT_THREADS / 16) {
      const int w = ob / V, v = ob % V;
      float p = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) p = fmaf(att[w * RV_U + 8 * sub + i], S.wfc[(8 * sub + i) * V + v], p);
      p = row16_sum(p) + S.wfc[RV_U * V + v];
      if (sub == 0) {
        lg[w * RV_MAX_VOCAB + v] = p;
        if (d.step_logits) d.step_logits[(((size_t)step * d.B + b) * W + w) * V + v] = p;
      }
    }
  }
  __syncthreads();
  RV_STAMP(d, step, 6);
  if (d.dbg_stop == 6) return;
  }   // !all_fin

  // G: sampler / beam step, wave 0: lane = candidate (beam w, token v), W*V <= 64
  if (tid < 64) {
    const int lane = tid, w = lane / V, v = lane % V;
    const bool cand = lane < W * V;
    const size_t o = ((size_t)step * d.B + b) * W;
    float val = -INFINITY;
    if (d.greedy) {
      if (cand) val = lg[v];
    } else if (cand) {
      float m = lg[w * RV_MAX_VOCAB];
      for (int x = 1; x < V; ++x) m = fmaxf(m, lg[w * RV_MAX_VOCAB + x]);
      float ssum = 0.f;
      for (int x = 0; x < V; ++x) ssum += __expf(lg[w * RV_MAX_VOCAB + x] - m);
      const float lse = __logf(ssum);
      const bool fin = S.fin[w] != 0;
      const float lp = fin ? (v == d.end_token ? 0.f : -FLT_MAX) : (lg[w * RV_MAX_VOCAB + v] - m) - lse;
      val = S.lprob[w] + lp;
    }
    // top-W by repeated wave max; ties -> lowest flat index (tf.math.top_k order)
    bool taken = !cand;
    int my_word = 0, my_par = 0; float my_val = 0.f;
    for (int k = 0; k < W; ++k) {
      const float mx = wave_max_fast(taken ? -INFINITY : val);
      const unsigned long long hit = __ballot(!taken && (val == mx || mx == -INFINITY));
      const int win = __ffsll((long long)hit) - 1;
      const float wval = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(val), win));
      if (lane == win) taken = true;
      if (lane == k) { my_word = win % V; my_par = win / V; my_val = wval; }
    }
    bool nf = false; int nl = 0;
    if (lane < W) {
      if (d.greedy) {
        nf = S.fin[0] != 0 || my_word == d.end_token;
      } else {
        const bool pf = S.fin[my_par] != 0;
        nf = pf || my_word == d.end_token;
        nl = S.len[my_par] + (pf ? 0 : 1);
      }
    }
    const unsigned long long fmask = __ballot(lane < W && nf);
    if (lane < W) {
      d.step_ids[o + lane] = my_word; d.parent_ids[o + lane] = my_par; d.step_scores[o + lane] = my_val;
      d.tok[row0 + lane] = my_word;
      d.finished[row0 + lane] = nf;
      if (!d.greedy) { d.log_probs[row0 + lane] = my_val; d.lengths[row0 + lane] = nl; }
      S.parent[lane] = my_par;
    }
    if (lane == 0 && __popcll(fmask) == W) atomicAdd(&d.nfin[step], 1);
  }
  __syncthreads();
  RV_STAMP(d, step, 7);
  if (d.dbg_stop == 7) return;

  // H: next-step state of every stacked cell, gathered by parent beam: layer 0 input = [attention | h_0],
  //    layer k input's own half = h_k, cell states c_k.  The top layer's h / c are already in LDS.
  const int top = d.depth - 1;
  for (int i = tid; i < W * RV_U; i += ATT_THREADS) {
    const int w = i >> 7, e = i & 127, p = S.parent[w];
    d.xh[(row0 + w) * RV_E + e] = att[p * RV_U + e];
    (d.xh + top * d.ls_xh)[(row0 + w) * RV_E + RV_U + e] = q[p * RV_U + e];
    (d.c + top * d.ls_c)[(row0 + w) * RV_U + e] = S.cnew[p * RV_U + e];
    for (int k = 0; k < top; ++k) {
      (d.xh + k * d.ls_xh)[(row0 + w) * RV_E + RV_U + e] = (d.h_new + k * d.ls_c)[(row0 + p) * RV_U + e];
      (d.c + k * d.ls_c)[(row0 + w) * RV_U + e] = (d.c_new + k * d.ls_c)[(row0 + p) * RV_U + e];
    }
  }
}

// =====================================================================================
// Two-pass attend (exact reference dataflow: scores from keys, then context from values).
// Used for Bahdanau attention and whenever per-step alignments are tapped.
// TB / TD: compile-time trip counts of the score / context sweeps (T_m <= 32*TB and <= 8*TD), so every
// load of a sweep is issued before its first use.
template <int W, int TB, int TD>
__global__ __launch_bounds__(ATT_THREADS) void k_dec_attend(DecState d, int step) {
  extern __shared__ __align__(16) float dsm[];
  const int Tm = d.Tm;
  const int TmP = (Tm + 3) & ~3;
  const AttLds L(W, TmP, false);
  float* q = dsm + L.q;        // [W][128]
  float* pq = dsm + L.pq;      // [W][128]
  float* sc = dsm + L.sc;      // [W][TmP]
  float* al = dsm + L.al;      // [TmP][WB]
  float* part = dsm + L.part;
  float* hcT = dsm + L.hcT;    // [384][WB]
  float* att = dsm + L.att;    // [W][128]
  float* lg = dsm + L.lg;      // [WB][8]
  __shared__ AttShared S;

  const int b = blockIdx.x, tid = threadIdx.x;
  if (step > 0 && d.nfin[step - 1] >= d.B) {      // whole batch finished earlier: propagate
    if (tid == 0) atomicAdd(&d.nfin[step], 1);
    return;
  }
  const size_t row0 = (size_t)b * W;
  att_prologue<W, 512>(d, S, q, hcT, row0, tid);
  __syncthreads();
  if (d.attention == 1) {   // Bahdanau: processed query = q . W_q ; thread = (column, K quarter)
    const int jj = tid & 127, kq = tid >> 7;
    float acc[W];
#pragma unroll
    for (int w = 0; w < W; ++w) acc[w] = 0.f;
    for (int k = kq * 32; k < kq * 32 + 32; ++k) {
      const float wq = d.W_q[k * RV_U + jj];
#pragma unroll
      for (int w = 0; w < W; ++w) acc[w] = fmaf(q[w * RV_U + k], wq, acc[w]);
    }
#pragma unroll
    for (int w = 0; w < W; ++w) part[(kq * W + w) * RV_U + jj] = acc[w];
    __syncthreads();
    if (tid < RV_U)
#pragma unroll
      for (int w = 0; w < W; ++w)
        pq[w * RV_U + tid] = (part[(0 * W + w) * RV_U + tid] + part[(1 * W + w) * RV_U + tid]) +
                             (part[(2 * W + w) * RV_U + tid] + part[(3 * W + w) * RV_U + tid]);
    __syncthreads();
  }

  // ---- B: scores.  A DPP row (16 lanes) shares one memory step; all key loads in flight first.
  {
    const int sub = tid & 15, grp = tid >> 4;
    float qr[W][8], vr[8];
    const float* qsrc = d.attention == 1 ? pq : q;
#pragma unroll
    for (int w = 0; w < W; ++w)
#pragma unroll
      for (int i = 0; i < 8; ++i) qr[w][i] = qsrc[w * RV_U + 8 * sub + i];
#pragma unroll
    for (int i = 0; i < 8; ++i) vr[i] = d.attention == 1 ? d.v_att[8 * sub + i] : 0.f;
    const float* kbase = d.keys + (size_t)b * Tm * RV_U + 8 * sub;
    const uint8_t* mrow = d.mask + (size_t)b * Tm;
    float4 k0[TB], k1[TB];
    uint8_t mk[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      const int t = min(grp + 32 * u, Tm - 1);
      const float4* kp = reinterpret_cast<const float4*>(kbase + (size_t)t * RV_U);
      k0[u] = kp[0]; k1[u] = kp[1];
      mk[u] = mrow[t];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      const int t = grp + 32 * u;
      const float kk[8] = {k0[u].x, k0[u].y, k0[u].z, k0[u].w, k1[u].x, k1[u].y, k1[u].z, k1[u].w};
      float mine = 0.f;
#pragma unroll
      for (int w = 0; w < W; ++w) {
        float p = 0.f;
        if (d.attention == 1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) p = fmaf(vr[i], tanhf(kk[i] + qr[w][i]), p);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) p = fmaf(kk[i], qr[w][i], p);
        }
        p = row16_sum(p);
        mine = sub == w ? p : mine;
      }
      if (t < Tm && sub < W) sc[sub * TmP + t] = mk[u] ? mine : -INFINITY;   // _maybe_mask_score
    }
  }

  // ---- D (loads): every value row this thread needs, issued before the softmax so the memory
  //         latency hides under phase C.  thread = (4 columns, 1 of 8 interleaved t-groups)
  const int cg = tid & 63, tg = tid >> 6;
  float4 vv[TD];
  {
    const float* vbase = d.values + (size_t)b * Tm * RV_E + 4 * cg;
#pragma unroll
    for (int u = 0; u < TD; ++u) {
      const int t = min(tg + 8 * u, Tm - 1);
      vv[u] = *reinterpret_cast<const float4*>(vbase + (size_t)t * RV_E);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();

  // ---- C: softmax over T_m, one wave per beam
  {
    const int lane = tid & 63, wv = tid >> 6;
    if (wv < W) {
      const int w = wv;
      float m = -INFINITY;
      for (int t = lane; t < Tm; t += 64) m = fmaxf(m, sc[w * TmP + t]);
      m = wave_max(m);
      float sum = 0.f;
      for (int t = lane; t < Tm; t += 64) {
        const float e = expf(sc[w * TmP + t] - m);
        sc[w * TmP + t] = e;
        sum += e;
      }
      sum = wave_sum(sum);
      for (int t = lane; t < Tm; t += 64) {
        const float a = sc[w * TmP + t] / sum;
        al[t * WB + w] = a;
        if (d.step_align) d.step_align[(((size_t)step * d.B + b) * W + w) * Tm + t] = a;
      }
    }
  }
  __syncthreads();

  // ---- D (math): context = sum_t alpha_t * values_t
  {
    f2 acc[W][2];
#pragma unroll
    for (int w = 0; w < W; ++w) { acc[w][0] = f2{0.f, 0.f}; acc[w][1] = f2{0.f, 0.f}; }
#pragma unroll
    for (int u = 0; u < TD; ++u) {
      const int t = tg + 8 * u;
      if (t < Tm) {
        float a[WB];
        *reinterpret_cast<float4*>(a) = *reinterpret_cast<const float4*>(&al[t * WB]);
        if (W > 4) *reinterpret_cast<float4*>(a + 4) = *reinterpret_cast<const float4*>(&al[t * WB + 4]);
#pragma unroll
        for (int w = 0; w < W; ++w) {
          acc[w][0] = __builtin_elementwise_fma(f2{a[w], a[w]}, f2{vv[u].x, vv[u].y}, acc[w][0]);
          acc[w][1] = __builtin_elementwise_fma(f2{a[w], a[w]}, f2{vv[u].z, vv[u].w}, acc[w][1]);
        }
      }
    }
#pragma unroll
    for (int w = 0; w < W; ++w)
      *reinterpret_cast<float4*>(&part[((tg * W + w) * RV_E) + 4 * cg]) =
          make_float4(acc[w][0].x, acc[w][0].y, acc[w][1].x, acc[w][1].y);
  }
  __syncthreads();
  for (int i = tid; i < W * RV_E; i += ATT_THREADS) {     // fixed-order reduction of the 8 partials
    const int w = i >> 8, col = i & 255;
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += part[(g * W + w) * RV_E + col];
    hcT[(RV_U + col) * WB + w] = s;
  }
  __syncthreads();
  att_tail<W, 512>(d, S, q, hcT, part, att, lg, b, step, tid);
}

// =====================================================================================
// Single-pass ("flash") attend for Luong attention -- the production path.
// The per-step attention is HBM-bound: keys [T_m,128] + values [T_m,256] per chunk are re-read every
// step (130 MB per step at B = 256).  Luong's score is linear in the keys, and keys = values . W_mem,
// so   score_t = q . (values_t . W_mem) = values_t . q'   with  q' = W_mem . q   (256-vector).
// One sweep over `values` then yields scores AND context (online softmax), and `keys` is never read:
// a third fewer bytes.  fp32 re-association only (|delta score| ~ 1e-6; tolerance 1e-4).
// 32 independent streams per chunk (a DPP row of 16 lanes each, 16 columns per lane) walk
// t = stream, stream+32, ... with a 3-deep register prefetch, keep their own running
// (max, sum, context[W][256]) and are merged once at the end in a fixed order.
// q' carries log2(e) so the exponentials are bare v_exp_f32.
// NT = 512: 32 streams, one workgroup per CU (slabs up to the CU count).  NT = 256: 16 streams, 71 KB of LDS
// and 196 VGPRs -> TWO workgroups per CU, so one chunk's latency-bound phases (cell-output fetch, q',
// attention layer, beam step) run under the other chunk's HBM-bound sweep (slabs larger than the CU count).
template <int W, int NT>
__global__ __launch_bounds__(NT) void k_dec_attend_flash(DecState d, const float* __restrict__ WmemT, int step) {
  constexpr int ATT_THREADS = NT, NW = NT / 64, NS = NT / 16;
  extern __shared__ __align__(16) float dsm[];
  const int Tm = d.Tm;
  const AttLds L(W, 0, true, NT);
  float* q = dsm + L.q;        // [W][128]
  float* qp = dsm + L.pq;      // [W][256]  q' * log2(e)
  float* ml = dsm + L.sc;      // [32][WB][2] per-stream (max, sum)
  float* mg = dsm + L.al;      // [WB][2] merged (max, 1/sum)
  float* part = dsm + L.part;  // [8][W][256]
  float* hcT = dsm + L.hcT;    // [384][WB]
  float* att = dsm + L.att;    // [W][128]
  float* lg = dsm + L.lg;      // [WB][8]
  float* fold = dsm + L.fold;
  __shared__ AttShared S;

  const int b = blockIdx.x, tid = threadIdx.x;
  if (step > 0 && d.nfin[step - 1] >= d.B) {
    if (tid == 0) atomicAdd(&d.nfin[step], 1);
    return;
  }
  const size_t row0 = (size_t)b * W;
  RV_STAMP(d, step, 0);
  const int lane = tid & 63, wv = tid >> 6, sub = lane & 15, sid = tid >> 4;   // sid: stream 0..NS-1
  constexpr int PD = 3;                                                       // prefetch depth (iterations)
  const float* vbase = d.values + (size_t)b * Tm * RV_E + 4 * sub;
  const uint8_t* mrow = d.mask + (size_t)b * Tm;
  const int nit = (Tm + NS - 1) / NS;
  float4 pv[PD][4];
  uint8_t pm[PD];
  auto issue = [&](int slot, int it) {
    const int t = min(sid + NS * it, Tm - 1);
    const float* p = vbase + (size_t)t * RV_E;
#pragma unroll
    for (int m = 0; m < 4; ++m) pv[slot][m] = *reinterpret_cast<const float4*>(p + 64 * m);
    pm[slot] = mrow[t];
  };
  // small loads first, INTO REGISTERS (vmcnt retires in issue order), then the stream, then the LDS
  // writes: the prologue waits one short round trip while the value rows are already in flight.
  constexpr int NC = (W * RV_U + ATT_THREADS - 1) / ATT_THREADS;
  constexpr int NWF = (RV_U * RV_MAX_VOCAB + ATT_THREADS - 1) / ATT_THREADS;   // W_fc staging slots per thread
  float hr[NC], cr[NC], wfr[NWF];
  int sm0 = 0, sm2 = 0; float sm1 = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int idx = tid + ATT_THREADS * i;
    hr[i] = idx < W * RV_U ? (d.h_new + (d.depth - 1) * d.ls_c)[row0 * RV_U + idx] : 0.f;
    cr[i] = idx < W * RV_U ? (d.c_new + (d.depth - 1) * d.ls_c)[row0 * RV_U + idx] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < NWF; ++i) { const int idx = tid + ATT_THREADS * i; wfr[i] = idx < RV_U * d.V ? d.W_fc[idx] : 0.f; }
  const float bfr = tid < d.V ? d.b_fc[tid] : 0.f;
  if (tid < W) { sm0 = d.finished[row0 + tid]; sm1 = d.log_probs[row0 + tid]; sm2 = d.lengths[row0 + tid]; }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < PD; ++k) issue(k, k);            // the stream starts before any math
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int idx = tid + ATT_THREADS * i;
    if (idx < W * RV_U) { q[idx] = hr[i]; hcT[(idx & 127) * WB + (idx >> 7)] = hr[i]; S.cnew[idx] = cr[i]; }
  }
#pragma unroll
  for (int i = 0; i < NWF; ++i) { const int idx = tid + ATT_THREADS * i; if (idx < RV_U * d.V) S.wfc[idx] = wfr[i]; }
  if (tid < d.V) S.wfc[RV_U * d.V + tid] = bfr;
  if (tid < W) { S.fin[tid] = sm0; S.lprob[tid] = sm1; S.len[tid] = sm2; }
  __syncthreads();
  RV_STAMP(d, step, 1);
  // A chunk whose W beams are ALL finished contributes only (beam, end-token) candidates at unchanged
  // scores (finished rows are replaced by [min.., 0 at '^']): its logits are never read,
  // so the cell output, attention and output layer are skipped and only the beam bookkeeping runs.  The
  // states written for such a chunk are never used for an output again.  (Not under debug taps, which
  // record logits of finished beams, and not for greedy decoding, whose rows keep sampling.)
  bool all_fin = !d.greedy && !d.step_logits && !d.step_align;
  for (int w = 0; w < W; ++w) all_fin = all_fin && S.fin[w] != 0;
  if (all_fin) {
    att_tail<W, NT>(d, S, q, hcT, part, att, lg, b, step, tid, true);
    return;
  }
  // q' = W_mem . q (times log2 e): thread = (4 columns, 1 of NW j-groups), W_memT is [128][256]
  {
    constexpr int JPG = RV_U / NW;                      // rows of W_memT per j-group (16 or 32)
    const int c4 = tid & 63, jg = tid >> 6;
    f2 acc[W][2];
#pragma unroll
    for (int w = 0; w < W; ++w) { acc[w][0] = f2{0.f, 0.f}; acc[w][1] = f2{0.f, 0.f}; }
#pragma unroll
    for (int j0 = 0; j0 < JPG; j0 += 16) {
      float4 wm[16];
      const float* wp = WmemT + (size_t)(JPG * jg + j0) * RV_E + 4 * c4;
#pragma unroll
      for (int u = 0; u < 16; ++u) wm[u] = *reinterpret_cast<const float4*>(wp + (size_t)u * RV_E);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        float hv[WB];
        *reinterpret_cast<float4*>(hv) = *reinterpret_cast<const float4*>(&hcT[(JPG * jg + j0 + u) * WB]);
        if (W > 4) *reinterpret_cast<float4*>(hv + 4) = *reinterpret_cast<const float4*>(&hcT[(JPG * jg + j0 + u) * WB + 4]);
#pragma unroll
        for (int w = 0; w < W; ++w) {
          acc[w][0] = __builtin_elementwise_fma(f2{hv[w], hv[w]}, f2{wm[u].x, wm[u].y}, acc[w][0]);
          acc[w][1] = __builtin_elementwise_fma(f2{hv[w], hv[w]}, f2{wm[u].z, wm[u].w}, acc[w][1]);
        }
      }
    }
#pragma unroll
    for (int w = 0; w < W; ++w)
      *reinterpret_cast<float4*>(&part[(jg * W + w) * RV_E + 4 * c4]) =
          make_float4(acc[w][0].x, acc[w][0].y, acc[w][1].x, acc[w][1].y);
  }
  __syncthreads();
  for (int i = tid; i < W * RV_E; i += ATT_THREADS) {
    const int w = i >> 8, col = i & 255;
    float s0 = 0.f;
#pragma unroll
    for (int g = 0; g < NW; ++g) s0 += part[(g * W + w) * RV_E + col];
    qp[i] = s0 * LOG2E;
  }
  __syncthreads();

  RV_STAMP(d, step, 2);
  // ---- the sweep.  Lane (sub) owns columns {4 sub + 64 m + 0..3 : m = 0..3} of its stream's rows.
  // Beam w's running (max, sum) of a stream lives in lane sub == w of its row; the rescale factor and
  // the new weight reach the other lanes through DPP row_newbcast.
  float M = -INFINITY, l = 0.f;
  f2 acc[W][8];
#pragma unroll
  for (int w = 0; w < W; ++w)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[w][i] = f2{0.f, 0.f};
  for (int it0 = 0; it0 < nit; it0 += PD) {
#pragma unroll
    for (int k = 0; k < PD; ++k) {
      const int it = it0 + k;
      if (it < nit) {                                   // wave-uniform
        asm volatile("" ::: "memory");                  // keep the q' LDS reads inside the loop (LICM would
                                                        // pin 16*W registers and spill; spill traffic drains vmcnt)
        const int t = sid + NS * it;
        const bool live = t < Tm && pm[k] != 0;
        f2 v[8];
#pragma unroll
        for (int m = 0; m < 4; ++m) { v[2 * m] = f2{pv[k][m].x, pv[k][m].y}; v[2 * m + 1] = f2{pv[k][m].z, pv[k][m].w}; }
        if (it + PD < nit) issue(k, it + PD);           // refill this slot (registers already copied)
        float mys = -INFINITY;
#pragma unroll
        for (int w = 0; w < W; ++w) {
          f2 pp = f2{0.f, 0.f};
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const float4 qv = *reinterpret_cast<const float4*>(&qp[w * RV_E + 4 * sub + 64 * m]);
            pp = __builtin_elementwise_fma(v[2 * m], f2{qv.x, qv.y}, pp);
            pp = __builtin_elementwise_fma(v[2 * m + 1], f2{qv.z, qv.w}, pp);
          }
          const float sw = row16_sum(pp.x + pp.y);      // log2-domain score, uniform across the row
          mys = sub == w ? sw : mys;
        }
        if (d.step_align && sub < W && t < Tm)          // tap: raw masked score (host normalises)
          d.step_align[(((size_t)step * d.B + b) * W + sub) * Tm + t] = live ? mys : -INFINITY;
        mys = live ? mys : -INFINITY;                   // _maybe_mask_score
        const float Mn = fmaxf(M, mys);
        const float mref = Mn == -INFINITY ? 0.f : Mn;
        const float scl = exp2f(M - mref), pe = exp2f(mys - mref);
        l = fmaf(l, scl, pe);
        M = Mn;
#pragma unroll
        for (int w = 0; w < W; ++w) {
          const float sc_w = row_bcast(scl, w), pe_w = row_bcast(pe, w);
#pragma unroll
          for (int i = 0; i < 8; ++i)
            acc[w][i] = __builtin_elementwise_fma(acc[w][i], f2{sc_w, sc_w}, f2{pe_w, pe_w} * v[i]);
        }
      }
    }
  }
  RV_STAMP(d, step, 3);
  // ---- merge the NS streams: global max / sum per beam, then a fixed-order sum of rescaled contexts
  if (sub < W) { ml[(sid * WB + sub) * 2] = M; ml[(sid * WB + sub) * 2 + 1] = l; }
  __syncthreads();
  for (int w = wv; w < W; w += NW) {     // a wave merges beam w: lane = stream
    const float Ms = lane < NS ? ml[(lane * WB + w) * 2] : -INFINITY;
    const float ls = lane < NS ? ml[(lane * WB + w) * 2 + 1] : 0.f;
    const float Mg = wave_max_fast(Ms);
    const float lsum = wave_sum_fast(Ms == -INFINITY ? 0.f : ls * exp2f(Ms - Mg));
    if (lane == 0) {
      mg[w * 2] = Mg;
      mg[w * 2 + 1] = 1.0f / lsum;       // all-masked chunk: 1/0 -> inf, context NaN like the reference's softmax
    }
  }
  __syncthreads();
  {
    const float Mg = sub < W ? mg[sub * 2] : 0.f;
    const float fs = M == -INFINITY ? 0.f : exp2f(M - Mg);          // this stream's weight for beam `sub`
    // fold the wave's 4 streams (its 4 DPP rows hold the same columns) through a wave-private 16-lane LDS
    // slab: row 3 -> row 1, row 2 -> row 0, row 1 -> row 0.  Same-wave exchange: no workgroup barrier.
    constexpr int SL = NT == 512 ? 32 : 16;             // slab lanes: 2-step fold where LDS allows, else 3-step
    float4* slab = reinterpret_cast<float4*>(fold + (size_t)wv * (W * 4 * SL * 4));
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const float f = row_bcast(fs, w);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[w][i] = acc[w][i] * f2{f, f};
    }
    const int row = lane >> 4;
    // both schedules add (row0 + row2) + (row1 + row3): identical results
    constexpr int NSTEP = NT == 512 ? 2 : 3;
#pragma unroll
    for (int stepf = 0; stepf < NSTEP; ++stepf) {
      bool is_src, is_dst; int sl;
      if (NT == 512) {       // rows {2,3} -> {0,1}, then row 1 -> row 0
        is_src = stepf == 0 ? lane >= 32 : row == 1;
        is_dst = stepf == 0 ? lane < 32 : row == 0;
        sl = stepf == 0 ? (lane & 31) : sub;
      } else {               // row 3 -> 1, row 2 -> 0, row 1 -> 0
        const int src = stepf == 0 ? 3 : (stepf == 1 ? 2 : 1), dst = stepf == 0 ? 1 : 0;
        is_src = row == src; is_dst = row == dst; sl = sub;
      }
      if (is_src) {
#pragma unroll
        for (int w = 0; w < W; ++w)
#pragma unroll
          for (int m = 0; m < 4; ++m)
            slab[(w * 4 + m) * SL + sl] = make_float4(acc[w][2 * m].x, acc[w][2 * m].y, acc[w][2 * m + 1].x, acc[w][2 * m + 1].y);
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): the wave's LDS writes have landed
      if (is_dst) {
#pragma unroll
        for (int w = 0; w < W; ++w)
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const float4 o = slab[(w * 4 + m) * SL + sl];
            acc[w][2 * m] += f2{o.x, o.y}; acc[w][2 * m + 1] += f2{o.z, o.w};
          }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);
    }
    if (row == 0) {
#pragma unroll
      for (int w = 0; w < W; ++w)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          *reinterpret_cast<float4*>(&part[(wv * W + w) * RV_E + 4 * sub + 64 * m]) =
              make_float4(acc[w][2 * m].x, acc[w][2 * m].y, acc[w][2 * m + 1].x, acc[w][2 * m + 1].y);
    }
  }
  __syncthreads();
  for (int i = tid; i < W * RV_E; i += ATT_THREADS) {     // fixed-order reduction over the waves
    const int w = i >> 8, col = i & 255;
    float s0 = 0.f;
#pragma unroll
    for (int g = 0; g < NW; ++g) s0 += part[(g * W + w) * RV_E + col];
    hcT[(RV_U + col) * WB + w] = s0 * mg[w * 2 + 1];
  }
  __syncthreads();
  RV_STAMP(d, step, 4);
  att_tail<W, NT>(d, S, q, hcT, part, att, lg, b, step, tid);
  RV_STAMP(d, step, 8);
}

// =====================================================================================
// Persistent decode: the WHOLE beam-search loop of one chunk in one workgroup, with the chunk's
// attention memory resident ON CHIP.
// The per-step attention is HBM-bound when `values` [T_m,256] fp32 (338 KB per chunk at T_m = 330)
// is re-streamed every step (k_dec_attend_flash: 86.6 MB per step at B = 256).  A CU's register file
// is 512 KB: thread (stream sid = tid>>4, sub = tid&15) loads rows t = sid + 32 it, columns
// {4 sub + 64 m + 0..3} of its chunk ONCE into NIT*16 VGPRs and keeps them for all L-1 steps.
// Chunks never interact inside the decode loop (the reference's loop only shares its stop test,
// "until every row of the slab is finished"): each workgroup runs until ITS beams are finished and
// records that step; k_dec_finalize extends earlier finishers exactly as the shared loop would
// (end token, unchanged score -- SURVEY.md A.5).  No hipGraph, no per-step launches, no HBM stream:
// per step the CU pulls only the weights (cell 512 KB + W_mem 128 KB + W_att 192 KB) from L2.
// Luong attention; beam search with W <= 8 (W <= 5 with two stacked cells: LDS) and greedy search.
// one decoder cell and W <= 5 leave 48 KB of LDS: 24 of the 256 weight rows the cell product streams from L2 every step stay on chip
__host__ __device__ constexpr bool persist_weight_cache(int W, int D) { return D == 1 && W <= 5; }
constexpr int PERSIST_MAX_NIT = 11;      // resident row groups of 32 steps: T_m <= 352
// `part` also holds the matrix-pipe attention's alignment image (two f16 parts of [32 NIT steps][8 beam slots] = NIT * 256 floats):
// with one beam that image is larger than the partial sums, so the block is sized for the larger of the two
__host__ __device__ constexpr int part_floats(int W) { return 4 * W * RV_G > PERSIST_MAX_NIT * 256 ? 4 * W * RV_G : PERSIST_MAX_NIT * 256; }
// ATT == 3 (matrix-pipe cell product): gate pre-activations z~ [W][RV_G + 16] (one plane: the product is complete inside a wave; rows
// padded so that the four beams a wave stores at once fall into different banks).  Weight cache in LDS: ALL 32 (k-step, gate) pairs
// of wave 0 (64 KB) -- wave 0 takes the output layer and the beam step first (4.7 k cycles) and would stream its share alone afterwards --
// and the last mxc_nc(W) pairs of each of the other seven waves (2 KB per pair and wave); the output layer's fragments (Wl16, 16 KB) sit in
// LDS as well.  This form needs neither `qp` / `att` (fp32 copies of h and ctx') nor `attT` / `hcT` nor the fp32 output layer in static LDS
__host__ __device__ constexpr int mxc_nc(int W) { return W <= 5 ? 2 : 1; }
constexpr int MXC_STATIC_LDS = 1024;     // static LDS of the matrix-pipe instantiations (bias, beam bookkeeping): the others keep ~8.7 KB of output layer
__host__ __device__ constexpr int mxc_cache_floats(int W) { return (32 + 7 * mxc_nc(W)) * 512; }
// two decoder cells on the matrix pipe: per step and wave 48 pairs of cell 0's product ([ctx' | h_1 | h_0] . [W_a ; A_h W_a ; U_0], K = 384) and
// 16 of cell 1's recurrent product (h_1 . U_1) at the end of the step, 16 of cell 1's input product (h_0 . W_1) after cell 0's gates.  The
// LDS that is left (W <= 5) keeps the first MXC2_CACHE of wave 0's 64 end-of-step pairs: wave 0 joins that stream 4.7 k cycles late
constexpr int MXC2_CACHE = 32;
constexpr int MXC_ZS = RV_G + 16;
__host__ __device__ constexpr int part_floats_mxc(int W) { return W * MXC_ZS > PERSIST_MAX_NIT * 256 ? W * MXC_ZS : PERSIST_MAX_NIT * 256; }
struct PersistLds {
  int attT, zb, cS, qp, part, ctxp, hcT, att, ml, mg, lg, fold, h0T, cS1, b1s, pq, vat, xim, wl16, wcache, total;
  // ATT: 0 Luong on fp32 rows, 1 Bahdanau, 2 Luong on the matrix pipe, 3 = 2 + the cell product on the matrix pipe,
  // 4 = Bahdanau scores on the VALU (fp32 key rows) with context, cell product and output layer on the matrix pipe as in 3
  __host__ __device__ PersistLds(int W, int D = 1, int ATT = 0) {
    const bool mxc = ATT == 3 || ATT == 4;
    int o = 0;
    attT = o; if (!mxc) o += RV_U * WB;   // attention vectors k-major beam-minor (cell input rows 0..127); mxc: the f16 image `xim` instead
    zb = o; o += RV_MAX_VOCAB * RV_G;  // one-hot token rows of the cell kernel + bias: [V][512]
    cS = o; o += 2 * W * RV_U;         // cell states, double-buffered: the new state of beam w comes from its parent's
    qp = o; if (!mxc) o += W * RV_U;   // h * log2(e): the score query (fp32 rows) and the output layer's h part
    part = o; o += mxc ? part_floats_mxc(W) : part_floats(W);     // cell-product partial sums [4][W][512] (end of step -> gates), then the attention
                                       // layer's h-part partial sums [16][W][128] (after the gates -> merge)
    ctxp = part;                       // context partial sums of the 8 waves [8][W][128]: one cell -> inside `part` (idle between
    if (D > 1 && !mxc) { ctxp = o; o += 8 * W * RV_U; }   // the gates and the end of the step); two cells -> own space (`part` holds h . A_h then)
    if (D > 1 && mxc) { ctxp = o; o += W * MXC_ZS; }      // two cells on the matrix pipe: h_1 . U_1 of the beams, [W][RV_G + 16] (`partU`)
    fold = o; o += mxc ? 1024 : 8 * 4 * 2 * 16 * 4;   // (mxc: only the query image, 2 parts x 1024 f16)   wave-private fold slab: 4 streams x 2 float4 x 16 lanes.  ctxp + fold also hold the
                                       // second cell's recurrent partial sums [3][W][512] between the end of a step and its gates
    hcT = o; if (!mxc) o += RV_U * WB;  // h of the top cell, k-major beam-minor (cell input rows 128..255, attention-layer input)
    att = o; if (!mxc) o += W * RV_U;
    ml = o; o += 64 * WB;              // per-stream max [32][WB], per-stream sum [32][WB]
    mg = o; o += 2 * WB;               // merged max, 1/sum
    lg = o; o += WB * RV_MAX_VOCAB;
    h0T = o; cS1 = o; b1s = o;
    if (D > 1) {                       // StackedRNNCells, second cell (basecaller.py:85-91)
      h0T = o; if (!mxc) o += RV_U * WB;   // h of cell 0, k-major beam-minor (input of cell 1 and rows 128..255 of cell 0's product); mxc: in `xim`
      cS1 = o; o += 2 * W * RV_U;
      b1s = o; o += RV_G;
    }
    pq = o; vat = o;
    if (ATT == 1 || ATT == 4) {        // Bahdanau (basecaller.py:131-132): processed query (h . W_q) per beam, and attention_v
      pq = o; o += W * RV_U;
      vat = o; o += RV_U;
    }
    xim = o;
    if (mxc) o += D > 1 ? 3072 : 2048; // [ctx' | h] ([ctx' | h_1 | h_0] with two cells) of the beams as MFMA A fragments: [32 | 48 k-blocks][16 rows][8] f16
    wl16 = o;
    if (mxc) o += 4096;                // the output layer [W_fc ; A_h W_fc] as B fragments (DecState::Wl16, 16 KB): wave 0's logits
    wcache = o;
    if (mxc) o += D > 1 ? MXC2_CACHE * 512 : mxc_cache_floats(W);
    else if (persist_weight_cache(W, D)) o += 24 * RV_G;   // 24 rows of the cell kernel kept in LDS (48 KB): the last 8 rows of K groups 1-3
    total = o;
  }
};

template <int W, int NIT, int D, int ATT = 0>
__global__ __launch_bounds__(512) void k_dec_persist(DecState d, const float* __restrict__ Wcat /*[256,512] = [W_in rows of the attention input ; U]*/,
                                                      const float* __restrict__ Wtok /*[V,512]*/, const float* __restrict__ bdec /*[512]*/,
                                                      const float* __restrict__ Wcat1 /*D == 2: [256,512] = [W_1 ; U_1]*/, const float* __restrict__ bdec1,
                                                      const float* __restrict__ Nh /*D == 1: A_h . W_fc [128,V]*/) {
  constexpr int NT = 512;
  constexpr bool BAH = ATT == 1 || ATT == 4;   // Bahdanau scores on the VALU (fp32 key rows)
  constexpr bool MX = ATT >= 2;            // the context on the matrix pipe (U' resident as f16 B fragments, alignments through an LDS image)
  constexpr bool MXS = ATT == 2 || ATT == 3;   // ... and the Luong scores (keys resident as f16 B fragments, query through an LDS image)
  constexpr bool MXC = ATT == 3 || ATT == 4;   // ... and the cell product [ctx' | h] . Wcat2 and the output layer (weights as B fragments: Wc16, Wl16)
  constexpr bool MXC2 = MXC && D == 2;     // two decoder cells, every product on the matrix pipe
  constexpr int ZS = MXC ? MXC_ZS : RV_G;  // row stride of the gate pre-activations in `part`
  constexpr int NC = MXC2 ? 0 : mxc_nc(W);
  constexpr int NR = (ATT == 3 && D == 1 && NIT <= 8) ? 5 : 0;   // T_m <= 256 leaves ~50 registers: waves 1-7 keep NR more (k-step, gate) pairs of their share there
  static_assert(!MX || D == 1 || ATT == 3, "two decoder cells run either on packed FMAs (ATT 0) or with everything on the matrix pipe (ATT 3)");
  static_assert(NIT <= PERSIST_MAX_NIT && NIT * 256 <= (MXC ? part_floats_mxc(W) : part_floats(W)), "the alignment image (2 f16 parts x 32 NIT steps x 8 slots) must fit `part`");
  static_assert(2 * 1024 * sizeof(_Float16) /* query image: 2 parts x [16 k-blocks][8 slots][8] f16 */ <= (8 * 4 * 2 * 16 * 4) * sizeof(float), "the query image must fit `fold`");
  extern __shared__ __align__(16) float dsm[];
  const PersistLds L(W, D, ATT);
  _Float16* xim = reinterpret_cast<_Float16*>(dsm + L.xim);   // MXC only
  float* pqs = dsm + L.pq;  float* vat = dsm + L.vat;   // ATT == 1 only
  constexpr bool CACHE = MXC || persist_weight_cache(W, D);
  float* wcache = dsm + L.wcache;                        // CACHE only: rows 72 g + 104 + i of Wcat at [(8 g + i) * 512], g = 0..2
  float* attT = dsm + L.attT;  float* zb = dsm + L.zb;  float* cS = dsm + L.cS;
  float* qp = dsm + L.qp;  float* part = dsm + L.part;  float* ctxp = dsm + L.ctxp;  float* hcT = dsm + L.hcT;  float* att = dsm + L.att;
  float* ml = dsm + L.ml;  float* mg = dsm + L.mg;  float* lg = dsm + L.lg;  float* fold = dsm + L.fold;
  float* h0T = dsm + L.h0T;  float* cS1 = dsm + L.cS1;  float* b1s = dsm + L.b1s;  float* partU = ctxp;   // D == 2 only (FMA form: spans ctxp + fold)
  // output layer weights, transposed to [v][k] (rows padded to 132 floats: the 8-lane groups of two outputs then read different banks)
  constexpr int FCW = RV_U + 4;
  // (the matrix-pipe form takes the output layer from the Wl16 fragments: it only needs the bias here, and neither `qp` nor `att`)
  __shared__ __align__(16) float s_wfc[(MXC ? 0 : RV_MAX_VOCAB * FCW) + RV_MAX_VOCAB];
  __shared__ __align__(16) float s_nh[(D == 1 && !MXC) ? RV_MAX_VOCAB * FCW : 4];
  constexpr int BFC = MXC ? 0 : RV_MAX_VOCAB * FCW;      // offset of b_fc in s_wfc
  __shared__ float s_lprob[WB];
  __shared__ int s_fin[WB], s_len[WB], s_parent[WB], s_tok[WB], s_allfin;
  __shared__ int s_qbig;                   // Bahdanau on the matrix pipe: some processed query of this step lies outside the fast form's range

  const int b = blockIdx.x, tid = threadIdx.x, Tm = d.Tm, V = d.V;
  const int lane = tid & 63, sub = lane & 15, sid = tid >> 4;
  const size_t row0 = (size_t)b * W;
  const int steps = d.L - 1;

  // ---- the chunk's attention memory -> registers (once).  Keys: the 16 lanes of a stream hold TWO rows at a time, lanes 0-7
  //      row sid + 32 (2p), lanes 8-15 row sid + 32 (2p + 1), 16 columns each, so that a score is an 8-lane reduction and one
  //      pass over the registers scores two rows.  U' (the context side): every lane 8 columns of each of the stream's rows.
  constexpr int NP = (NIT + 1) / 2;
  const int half = sub >> 3, s8 = sub & 7;
  float4 kr[MXS ? 1 : NP][4], ur[MX ? 1 : NIT][2];
  // MX: the same memory as MFMA B fragments (v_mfma_f32_16x16x32_f16), two f16 parts of the scaled values each.  Scores: wave
  // wv owns the 16-step tiles wv + 8 c; lane (n = step l % 16 of the tile, kq = l / 16) holds key[t][32 ks + 8 kq + 0..7].
  // Context: wave wv owns units 16 wv .. 16 wv + 15; lane (n = unit, kq) holds U'[32 ks + 8 kq + 0..7][unit].  The scales are
  // powers of two from the weights (|key| <= sum_k |W_mem[k][u]| because |enc_out| <= 1): nothing can overflow f16, and what
  // falls into its subnormals is below 2^-28 of the largest value the column can take.
  constexpr int NTT = (2 * NIT + 7) / 8;
  float4 kb[MXS ? NTT : 1][4][2], ub[MX ? NIT : 1][2];
  unsigned livebits = 0;
  if constexpr (MXS) {
    const float* cbase = d.values + (size_t)b * Tm * RV_E;
    const uint8_t* mrow = d.mask + (size_t)b * Tm;
    const int l16 = lane & 15, kq = lane >> 4, wv0 = tid >> 6;
#pragma unroll
    for (int c = 0; c < NTT; ++c) {
      const int t = 16 * (wv0 + 8 * c) + l16, tc = min(t, Tm - 1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        float v[8];
        *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(cbase + (size_t)tc * RV_E + 32 * ks + 8 * kq);
        *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(cbase + (size_t)tc * RV_E + 32 * ks + 8 * kq + 4);
        split_f16x8(v, d.mx_kscale, kb[c][ks][0], kb[c][ks][1]);
      }
      if (t < Tm && mrow[tc]) livebits |= 1u << c;          // _maybe_mask_score: padded steps never score
    }
  }
  if constexpr (MX) {
    const float* cbase = d.values + (size_t)b * Tm * RV_E;
    const int l16 = lane & 15, kq = lane >> 4, wv0 = tid >> 6;
#pragma unroll
    for (int ks = 0; ks < NIT; ++ks) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = cbase[(size_t)min(32 * ks + 8 * kq + j, Tm - 1) * RV_E + RV_U + 16 * wv0 + l16];
      split_f16x8(v, d.mx_uscale, ub[ks][0], ub[ks][1]);
    }
  }
  if constexpr (!MXS) {
    const float* cbase = d.values + (size_t)b * Tm * RV_E;
    const uint8_t* mrow = d.mask + (size_t)b * Tm;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int tc = min(sid + 32 * (2 * p + half), Tm - 1);
      const float* q = cbase + (size_t)tc * RV_E + 16 * s8;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        kr[p][m] = *reinterpret_cast<const float4*>(q + 4 * m);
        if (BAH) {   // Bahdanau: tanh(k + pq) = 1 - 2 / (1 + exp2((k + pq) * 2 log2 e)): the keys carry the factor from here on
          kr[p][m].x *= 2.0f * LOG2E; kr[p][m].y *= 2.0f * LOG2E; kr[p][m].z *= 2.0f * LOG2E; kr[p][m].w *= 2.0f * LOG2E;
        }
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int t = sid + 32 * it, tc = min(t, Tm - 1);
      if constexpr (!MX) {
        const float* q = cbase + (size_t)tc * RV_E + RV_U + 4 * sub;
#pragma unroll
        for (int m = 0; m < 2; ++m) ur[it][m] = *reinterpret_cast<const float4*>(q + 64 * m);
      }
      if (t < Tm && mrow[tc]) livebits |= 1u << it;       // _maybe_mask_score: padded steps never score
    }
  }
  // Bahdanau on the matrix pipe: tanh(k + q) = 1 - 2 / (1 + 2^(k' + q')) with k' = 2 log2(e) k, and 2^(k' + q') = 2^k' . 2^q': the resident
  // rows hold E_k = 2^k' (ONE exponential per key element for the whole decode), a step's queries come as E_q = 2^q', and a score element
  // costs a multiply-add and ONE reciprocal instead of an exponential and a reciprocal -- the loop is bound by the transcendental rate
  // (16 lanes per clock and CU).  Exact for |k'| <= 64 and |q'| <= 60 (the product stays a normal f32: |k| <= 22, |h . W_q| <= 20);
  // a chunk with a larger key keeps k' and takes the two-transcendental form for all its steps, a step with a larger query takes
  // 2^(log2(E_k) + q') -- both block-uniform choices, neither ever seen with weights of any plausible scale.
  bool kbig = false;
  if constexpr (BAH && MXC) {
    bool big = false;
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int m = 0; m < 4; ++m)
        big = big || !(fabsf(kr[p][m].x) <= 64.f && fabsf(kr[p][m].y) <= 64.f && fabsf(kr[p][m].z) <= 64.f && fabsf(kr[p][m].w) <= 64.f);
    kbig = __syncthreads_or(big ? 1 : 0) != 0;
    if (!kbig) {
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          kr[p][m].x = __builtin_amdgcn_exp2f(kr[p][m].x); kr[p][m].y = __builtin_amdgcn_exp2f(kr[p][m].y);
          kr[p][m].z = __builtin_amdgcn_exp2f(kr[p][m].z); kr[p][m].w = __builtin_amdgcn_exp2f(kr[p][m].w);
        }
    }
  }
  // ---- decoder initial state: zeros; start tokens; log_probs = [0, -inf, ...]
  // The cell's matrix-vector product of a step is taken on the PREVIOUS step's beams, before they are re-ordered (it only
  // needs their attention vector and h); the gate math then picks up the partial sums and the cell state of its parent
  // beam.  Zero initial state: partial sums 0, parents = identity.
  if (!MXC) {
    for (int i = tid; i < RV_U * WB; i += NT) attT[i] = 0.f;
    for (int i = tid; i < RV_U * WB; i += NT) hcT[i] = 0.f;
  }
  for (int i = tid; i < 2 * W * RV_U; i += NT) cS[i] = 0.f;
  for (int i = tid; i < (MXC ? W * ZS : 4 * W * RV_G); i += NT) part[i] = 0.f;
  if (MXC) for (int i = tid; i < (MXC2 ? 3072 : 2048); i += NT) dsm[L.xim + i] = 0.f;
  if (D > 1) {
    if (!MXC) for (int i = tid; i < RV_U * WB; i += NT) h0T[i] = 0.f;
    for (int i = tid; i < 2 * W * RV_U; i += NT) cS1[i] = 0.f;
    for (int i = tid; i < (MXC ? W * ZS : 3 * W * RV_G); i += NT) partU[i] = 0.f;
    b1s[tid] = bdec1[tid];
  }
  if (!MXC) {
    for (int i = tid; i < RV_U * V; i += NT) s_wfc[(i % V) * FCW + i / V] = d.W_fc[i];
    if (D == 1) for (int i = tid; i < RV_U * V; i += NT) s_nh[(i % V) * FCW + i / V] = Nh[i] * (1.0f / LOG2E);   // applied to qp = h * log2(e), the row-major copy of h
  }
  if (tid < V) s_wfc[BFC + tid] = d.b_fc[tid];
  if (MXC) {   // the output layer's fragments
    const uint4* src = reinterpret_cast<const uint4*>(d.Wl16);
    uint4* dst = reinterpret_cast<uint4*>(dsm + L.wl16);
    for (int i = tid; i < 1024; i += NT) dst[i] = src[i];
  }
  if (MXC2) {  // the first MXC2_CACHE of wave 0's end-of-step pairs (cell 0's product, image order)
    const uint4* src = reinterpret_cast<const uint4*>(d.Wc16);
    uint4* dst = reinterpret_cast<uint4*>(wcache);
    for (int i = tid; i < MXC2_CACHE * 128; i += NT) dst[i] = src[i];
  } else if (MXC) {   // wave 0's 32 pairs, then the last NC pairs of waves 1-7: 2 KB each, in image order
    const uint4* src = reinterpret_cast<const uint4*>(d.Wc16);
    uint4* dst = reinterpret_cast<uint4*>(wcache);
    for (int i = tid; i < 32 * 128; i += NT) dst[i] = src[i];
    if constexpr (NC > 0)
      for (int i = tid; i < 7 * NC * 128; i += NT) dst[32 * 128 + i] = src[(size_t)((1 + i / (NC * 128)) * 32 + 32 - NC) * 128 + i % (NC * 128)];
  } else if (CACHE) {
    for (int i = tid; i < 24 * (RV_G / 4); i += NT) {
      const int r = i >> 7, c = i & 127;
      *reinterpret_cast<float4*>(&wcache[r * RV_G + 4 * c]) =
          *reinterpret_cast<const float4*>(Wcat + (size_t)(72 * (r >> 3) + 104 + (r & 7)) * RV_G + 4 * c);
    }
  }
  if (BAH && tid < RV_U) vat[tid] = d.v_att[tid] * (-2.0f * LOG2E);   // score = sum_j v_j tanh(.) = const - 2 sum_j v_j / (1 + exp(.)): softmax drops the constant
  if (tid < WB) {
    s_tok[tid] = d.start_token; s_lprob[tid] = tid == 0 ? 0.f : -INFINITY;
    s_fin[tid] = 0; s_len[tid] = 0; s_parent[tid] = tid;
  }
  if (tid == 0) s_allfin = 0;
  for (int i = tid; i < V * RV_G; i += NT) zb[i] = Wtok[i] + bdec[i & (RV_G - 1)];   // W_dec[one_hot(v)] + b, resident
  __syncthreads();

  uint4 rbh[NR > 0 ? NR : 1], rbl[NR > 0 ? NR : 1];       // pairs 32 - NC - NR .. 32 - NC - 1 of this wave, resident for the whole decode
  if constexpr (NR > 0) {
    const uint4* wimg = reinterpret_cast<const uint4*>(d.Wc16) + (size_t)(tid >> 6) * (32 * 128) + (tid & 63);
#pragma unroll
    for (int r = 0; r < NR; ++r) { rbh[r] = wimg[(2 * (32 - NC - NR + r)) * 64]; rbl[r] = wimg[(2 * (32 - NC - NR + r) + 1) * 64]; }
  }
  int done_steps = steps;
  const int tid0 = tid;
  for (int step = 0; step < steps; ++step) {
    // Thread coordinates are re-derived from an opaque copy each step: otherwise every LDS / global address
    // of every phase is hoisted out of the loop and the hoisted values spill (the resident rows leave ~70 VGPRs).
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wv = tid >> 6, sub = lane & 15, sid = tid >> 4, row = lane >> 4;
    RV_STAMP(d, step, 0);
    if (BAH && MXC && tid == 0) s_qbig = 0;              // (set again, if at all, after the gates' barrier; last read before this step's last barrier)
    // ================= gates of this step: the cell product was taken at the end of the previous step on the parent beams
    const int cb = step & 1;                             // cell-state buffer holding the previous step's states
    // The chunk's resident rows are [keys | U'] = values . [W_mem | A_c] (A_c = the attention layer's context rows), built once
    // per slab by a GEMM: score_t = keys_t . h as in the reference (128-wide, no q' = W_mem h product per step), and the
    // attention vector = h . A_h + ctx' with ctx' = sum_t alpha_t U'_t (the context never has to pass through the attention
    // layer).  With ONE decoder cell the h part is folded into the weights at load time as well: the cell product of the
    // next step is [ctx' | h] . [W_a ; U + A_h W_a] and the logits are ctx' . W_fc + h . (A_h W_fc) + b, so no attention-layer
    // product is left in the step.  With two cells (cell 0 would need h_1 as a third input block) h . A_h stays explicit.
    float4 pw[8];                      // D == 2: this thread's whole slice of A_h = W_att[0:128] (8 rows x 4 columns), requested before the gate math
    auto ah_prefetch = [&]() {
      const float* wa = d.W_att + (size_t)(8 * (tid >> 5)) * RV_U + 4 * (tid & 31);
#pragma unroll
      for (int u = 0; u < 8; ++u) pw[u] = *reinterpret_cast<const float4*>(wa + (size_t)u * RV_U);
    };
    auto wq_prefetch = [&]() {           // ATT: this thread's slice of W_q (8 rows x 4 columns), requested before the gate math
      const float* wq = d.W_q + (size_t)(8 * (tid >> 5)) * RV_U + 4 * (tid & 31);
#pragma unroll
      for (int u = 0; u < 8; ++u) pw[u] = *reinterpret_cast<const float4*>(wq + (size_t)u * RV_U);
    };
    // Bahdanau + matrix pipe: this wave's B fragments of W_q (its 16 columns, K = 128: 4 k-steps x 2 parts), requested before the gate math
    uint4 wqf[(BAH && MXC) ? 8 : 1];
    if constexpr (BAH && MXC) {
      const uint4* wq = reinterpret_cast<const uint4*>(d.Wq16) + (size_t)wv * (8 * 64) + lane;
#pragma unroll
      for (int j = 0; j < 8; ++j) wqf[j] = wq[j * 64];
    } else if (BAH) wq_prefetch();
    if constexpr (MXC) {
      // W * 128 (beam, unit) items on 512 threads: waves 0-1 take two (W = 5).  Both items' operands are read first, then both are
      // computed, then stored: the second item rides in the first one's LDS and transcendental latencies instead of doubling the phase.
      // CELL 0: z = the end-of-step product of the PARENT beam + the token's row; CELL 1 (two cells): z = h_0 . W_1 of this beam (taken
      // right before) + h_1 . U_1 of the parent beam (end of the previous step) + b_1.  h goes to its segment of the A-fragment image:
      // one cell [ctx' | h]; two cells [ctx' | h_1 | h_0]; the top cell's h is the score query as well.
      auto mxc_gates = [&](auto cell_tag) {
        constexpr int CELL = decltype(cell_tag)::value;
        constexpr bool TOP = CELL == D - 1;
        constexpr int NITEM = (W * RV_U + NT - 1) / NT;
        float* cst = CELL == 0 ? cS : cS1;
        float z4[NITEM][4], cp[NITEM];
        bool ok[NITEM];
#pragma unroll
        for (int it = 0; it < NITEM; ++it) {
          const int idx = tid + it * NT;
          ok[it] = idx < W * RV_U;                             // wave-uniform (128 items per beam, 64 lanes per wave)
          const int w = ok[it] ? idx >> 7 : 0, u = idx & 127, pb = s_parent[w], tk = s_tok[w];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if constexpr (CELL == 0) z4[it][g] = part[pb * ZS + g * RV_U + u] + zb[tk * RV_G + g * RV_U + u];
            else z4[it][g] = (part[w * ZS + g * RV_U + u] + partU[pb * ZS + g * RV_U + u]) + b1s[g * RV_U + u];
          }
          cp[it] = cst[cb * W * RV_U + pb * RV_U + u];
        }
#pragma unroll
        for (int it = 0; it < NITEM; ++it) {
          if (!ok[it]) continue;
          const int idx = tid + it * NT, w = idx >> 7, u = idx & 127;
          const float c2 = fmaf(rv_sigmoid(z4[it][1]), cp[it], rv_sigmoid(z4[it][0]) * rv_tanh(z4[it][2]));
          const float hh = rv_sigmoid(z4[it][3]) * rv_tanh(c2);
          cst[(cb ^ 1) * W * RV_U + idx] = c2;
          {   // h as A fragments: h 2^14 in two f16 parts, k = 128 + u (top cell), 256 + u (cell 0 of two)
            const float sv = hh * 16384.f;
            const _Float16 hi = (_Float16)sv, lo = (_Float16)(sv - (float)hi);
            _Float16* xq = xim + ((((TOP ? 1 : 2) * RV_U + u) >> 3) * 16 + mx_row(w)) * 8 + (u & 7);
            xq[0] = hi; xq[8] = lo;
          }
          if constexpr (MXS && TOP) {   // the score query as MFMA A fragments: [k-block u / 8][row mx_row(w) (+ 1: low part)][u % 8] f16 of h log2(e) 2^14
            const float sv = (hh * LOG2E) * 16384.f;
            const _Float16 hi = (_Float16)sv, lo = (_Float16)(sv - (float)hi);
            _Float16* qa = reinterpret_cast<_Float16*>(fold) + ((u >> 3) * 16 + mx_row(w)) * 8 + (u & 7);
            qa[0] = hi; qa[8] = lo;
          }
        }
      };
      mxc_gates(std::integral_constant<int, 0>{});
      if constexpr (MXC2) {
        __syncthreads();
        // ---- cell 1's input product z_1a = h_0(new) . W_1 on the matrix pipe: every wave its 16 units x 4 gates, K = 128 (the h_0 segment
        //      of the image: k-steps 8..11), 16 pairs streamed from the W1c16 image through a window of 4; `part` (cell 0's z~, consumed by
        //      the gates above) takes the result, rows = this step's beams
        {
          const int l16 = lane & 15, kq = lane >> 4;
          const uint4* wimg = reinterpret_cast<const uint4*>(d.W1c16) + (size_t)wv * (32 * 128) + lane;
          const _Float16* xa = xim + (kq * 16 + l16) * 8;
          f4v acc[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = f4v{0.f, 0.f, 0.f, 0.f};
          uint4 bh[4], bl[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) { bh[i] = wimg[(2 * i) * 64]; bl[i] = wimg[(2 * i + 1) * 64]; }
#pragma unroll
          for (int p = 0; p < 16; ++p) {
            const h8 a = *reinterpret_cast<const h8*>(xa + (8 + (p >> 2)) * 512);
            acc[p & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, bl[p & 3]), acc[p & 3], 0, 0, 0);
            acc[p & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, bh[p & 3]), acc[p & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (p + 4 < 16) { bh[p & 3] = wimg[(2 * (p + 4)) * 64]; bl[p & 3] = wimg[(2 * (p + 4) + 1) * 64]; }
            __builtin_amdgcn_sched_barrier(0);
          }
          constexpr int NI = W > 4 ? 2 : 1;
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < NI; ++i)
              if (kq + 4 * i < W) part[(kq + 4 * i) * ZS + RV_U * g + 16 * wv + l16] = (acc[g][2 * i] + acc[g][2 * i + 1]) * d.mx_c1descale;
        }
        __syncthreads();
        mxc_gates(std::integral_constant<int, 1>{});
      }
    } else
    for (int idx = tid; idx < W * RV_U; idx += NT) {       // K-group sums in fixed order, gate math, cell update
      const int w = idx >> 7, u = idx & 127, pb = s_parent[w];
      float z4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = g * RV_U + u;
        z4[g] = (((part[(0 * W + pb) * RV_G + col] + part[(1 * W + pb) * RV_G + col]) + part[(2 * W + pb) * RV_G + col]) +
                   part[(3 * W + pb) * RV_G + col]) + zb[s_tok[w] * RV_G + col];
      }
      const float c2 = fmaf(rv_sigmoid(z4[1]), cS[cb * W * RV_U + pb * RV_U + u], rv_sigmoid(z4[0]) * rv_tanh(z4[2]));
      const float hh = rv_sigmoid(z4[3]) * rv_tanh(c2);
      cS[(cb ^ 1) * W * RV_U + idx] = c2;
      if (D > 1) h0T[u * WB + w] = hh; else { hcT[u * WB + w] = hh; qp[idx] = hh * LOG2E; }
      if constexpr (MXS) {     // the score query as MFMA A fragments: [k-block u / 8][row mx_row(w) (+ 1: low part)][u % 8] f16 of h log2(e) 2^14
        const float sv = (hh * LOG2E) * 16384.f;
        const _Float16 hi = (_Float16)sv, lo = (_Float16)(sv - (float)hi);
        _Float16* qa = reinterpret_cast<_Float16*>(fold) + ((u >> 3) * 16 + mx_row(w)) * 8 + (u & 7);
        qa[0] = hi; qa[8] = lo;
      }
    }
    __syncthreads();
    if (D > 1 && !MXC) {
      // ---- second cell: z_1 = h_0(new) . W_1 + [h_1(prev) . U_1 of the parent beam, taken at the end of the previous step] + b_1
      {
        const int c4 = tid & 127, kg = tid >> 7;           // 32 rows of W_1 per K group
        f2 acc[W][2];
#pragma unroll
        for (int w = 0; w < W; ++w) { acc[w][0] = f2{0.f, 0.f}; acc[w][1] = f2{0.f, 0.f}; }
        const float* wc = Wcat1 + (size_t)(32 * kg) * RV_G + 4 * c4;
        const float* xk = h0T + (32 * kg) * WB;
#pragma unroll 1
        for (int k0 = 0; k0 < 32; k0 += 8) {
          float4 wr[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) wr[i] = *reinterpret_cast<const float4*>(wc + (size_t)(k0 + i) * RV_G);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float xv[WB];
            *reinterpret_cast<float4*>(xv) = *reinterpret_cast<const float4*>(&xk[(k0 + i) * WB]);
            if (W > 4) *reinterpret_cast<float4*>(xv + 4) = *reinterpret_cast<const float4*>(&xk[(k0 + i) * WB + 4]);
#pragma unroll
            for (int w = 0; w < W; ++w) {
              acc[w][0] = __builtin_elementwise_fma(f2{xv[w], xv[w]}, f2{wr[i].x, wr[i].y}, acc[w][0]);
              acc[w][1] = __builtin_elementwise_fma(f2{xv[w], xv[w]}, f2{wr[i].z, wr[i].w}, acc[w][1]);
            }
          }
        }
#pragma unroll
        for (int w = 0; w < W; ++w)
          *reinterpret_cast<float4*>(&part[(kg * W + w) * RV_G + 4 * c4]) = make_float4(acc[w][0].x, acc[w][0].y, acc[w][1].x, acc[w][1].y);
      }
      ah_prefetch();
      __syncthreads();
      for (int idx = tid; idx < W * RV_U; idx += NT) {
        const int w = idx >> 7, u = idx & 127, pb = s_parent[w];
        float z4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = g * RV_U + u;
          const float zin = ((part[(0 * W + w) * RV_G + col] + part[(1 * W + w) * RV_G + col]) + part[(2 * W + w) * RV_G + col]) +
                            part[(3 * W + w) * RV_G + col];
          const float zrec = (partU[(0 * W + pb) * RV_G + col] + partU[(1 * W + pb) * RV_G + col]) + partU[(2 * W + pb) * RV_G + col];
          z4[g] = (zin + zrec) + b1s[col];
        }
        const float c2 = fmaf(rv_sigmoid(z4[1]), cS1[cb * W * RV_U + pb * RV_U + u], rv_sigmoid(z4[0]) * rv_tanh(z4[2]));
        const float hh = rv_sigmoid(z4[3]) * rv_tanh(c2);
        cS1[(cb ^ 1) * W * RV_U + idx] = c2; hcT[u * WB + w] = hh; qp[idx] = hh * LOG2E;
      }
      __syncthreads();
    }
    if constexpr (BAH && MXC) {
      // ================= Bahdanau: processed query pq = h . W_q (BahdanauAttention.query_layer, no bias) on the matrix pipe: A = the h rows
      //   of the [ctx' | h] image (k-steps 4..7), B = this wave's 16 columns of W_q (Wq16, requested before the gate math): complete in the wave
      const int l16 = lane & 15, kq = lane >> 4;
      const _Float16* xa = xim + (kq * 16 + l16) * 8;
      f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const h8 a = *reinterpret_cast<const h8*>(xa + (4 + ks) * 512);
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, wqf[2 * ks]), a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, wqf[2 * ks + 1]), a1, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < (W > 4 ? 2 : 1); ++i)
        if (kq + 4 * i < W) {
          const float qv = ((a0[2 * i] + a0[2 * i + 1]) + (a1[2 * i] + a1[2 * i + 1])) * (d.mx_qdescale * 2.0f * LOG2E);
          pqs[(kq + 4 * i) * RV_U + 16 * wv + l16] = qv;                       // q' ...
          fold[(kq + 4 * i) * RV_U + 16 * wv + l16] = __builtin_amdgcn_exp2f(fminf(fmaxf(qv, -126.f), 126.f));   // ... and E_q = 2^q' (`fold` holds no query image here)
          if (!(fabsf(qv) <= 60.f)) s_qbig = 1;
        }
      __syncthreads();
    } else if (BAH) {
      // ================= Bahdanau: processed query pq = h . W_q (BahdanauAttention.query_layer, no bias); thread = (4 columns,
      //   1 of 16 K groups of 8 rows), partial sums through `part` (free between the gates and the context phase)
      const int d4 = tid & 31, kg = tid >> 5;
      f2 acc[W][2];
#pragma unroll
      for (int w = 0; w < W; ++w) { acc[w][0] = f2{0.f, 0.f}; acc[w][1] = f2{0.f, 0.f}; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float hv[WB];
        *reinterpret_cast<float4*>(hv) = *reinterpret_cast<const float4*>(&hcT[(8 * kg + u) * WB]);
        if (W > 4) *reinterpret_cast<float4*>(hv + 4) = *reinterpret_cast<const float4*>(&hcT[(8 * kg + u) * WB + 4]);
#pragma unroll
        for (int w = 0; w < W; ++w) {
          acc[w][0] = __builtin_elementwise_fma(f2{hv[w], hv[w]}, f2{pw[u].x, pw[u].y}, acc[w][0]);
          acc[w][1] = __builtin_elementwise_fma(f2{hv[w], hv[w]}, f2{pw[u].z, pw[u].w}, acc[w][1]);
        }
      }
#pragma unroll
      for (int w = 0; w < W; ++w)
        *reinterpret_cast<float4*>(&part[(kg * W + w) * RV_U + 4 * d4]) =
            make_float4(acc[w][0].x, acc[w][0].y, acc[w][1].x, acc[w][1].y);
      __syncthreads();
      for (int i = tid; i < W * RV_U; i += NT) {
        const int w = i >> 7, col = i & 127;
        float s0 = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) s0 += part[(g * W + w) * RV_U + col];
        pqs[i] = s0 * (2.0f * LOG2E);
      }
      __syncthreads();
    }
    RV_STAMP(d, step, 2);
    // ================= (two cells) attention layer, h part: h . A_h ; thread = (4 columns, 1 of 16 K groups of 8 rows), one batch
    if (D > 1 && !MXC) {
      const int d4 = tid & 31, kg = tid >> 5;
      f2 acc[W][2];
#pragma unroll
      for (int w = 0; w < W; ++w) { acc[w][0] = f2{0.f, 0.f}; acc[w][1] = f2{0.f, 0.f}; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float hv[WB];
        *reinterpret_cast<float4*>(hv) = *reinterpret_cast<const float4*>(&hcT[(8 * kg + u) * WB]);
        if (W > 4) *reinterpret_cast<float4*>(hv + 4) = *reinterpret_cast<const float4*>(&hcT[(8 * kg + u) * WB + 4]);
#pragma unroll
        for (int w = 0; w < W; ++w) {
          acc[w][0] = __builtin_elementwise_fma(f2{hv[w], hv[w]}, f2{pw[u].x, pw[u].y}, acc[w][0]);
          acc[w][1] = __builtin_elementwise_fma(f2{hv[w], hv[w]}, f2{pw[u].z, pw[u].w}, acc[w][1]);
        }
      }
#pragma unroll
      for (int w = 0; w < W; ++w)    // (the cell-product partial sums in `part` were consumed by the gates above)
        *reinterpret_cast<float4*>(&part[(kg * W + w) * RV_U + 4 * d4]) =
            make_float4(acc[w][0].x, acc[w][0].y, acc[w][1].x, acc[w][1].y);
    }
    RV_STAMP(d, step, 3);

    if constexpr (MXS) {
      // ================= scores on the matrix pipe: rows = beams, columns = the 16 steps of a tile, K = 128 units.
      //   C/D map: lane l holds column l % 16 and rows 4 g + i (g = l / 16).  Beam w sits in rows mx_row(w), + 1 (its high and its low
      //   part: see mx_row), so that registers 0 + 1 hold beams 0-3 (one per 16-lane group, all 64 lanes busy) and 2 + 3 beams 4-7: the
      //   softmax below touches W <= 4 ? 1 : 2 values per tile.  A row of A only feeds its own row of C, so the rows of absent beams
      //   hold whatever the LDS image holds there, harmlessly.
      constexpr int NI = W > 4 ? 2 : 1;
      const int l16 = lane & 15, kq = lane >> 4;
      float sc[NTT][NI];
      {
        const _Float16* qa = reinterpret_cast<const _Float16*>(fold) + (kq * 16 + l16) * 8;    // row l16 of k-block 4 ks + kq
        // the tiles in turns (against the low, then against the high part of the keys): an MFMA accumulates onto a result that is NTT
        // instructions old, not onto the one issued just before it
        f4v acc[NTT];
#pragma unroll
        for (int c = 0; c < NTT; ++c) acc[c] = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {                    // one k-step of the query at a time: 4 registers of A fragments live
          const h8 a = *reinterpret_cast<const h8*>(qa + ks * 512);
#pragma unroll
          for (int c = 0; c < NTT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, kb[c][ks][1]), acc[c], 0, 0, 0);
#pragma unroll
          for (int c = 0; c < NTT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, kb[c][ks][0]), acc[c], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < NTT; ++c) {
          const bool live = (livebits >> c) & 1u;
#pragma unroll
          for (int i = 0; i < NI; ++i)
            sc[c][i] = (live && kq + 4 * i < W) ? (acc[c][2 * i] + acc[c][2 * i + 1]) * d.mx_kdescale : -INFINITY;
        }
      }
      RV_STAMP(d, step, 4);
      // ================= softmax over the chunk's T_m steps: per-wave (max, sum) of every beam, one fixed-order merge
      //   (raw v_exp_f32: arguments <= 0, a result below 2^-126 is an alignment of 0 either way)
      float mrow_[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        float m = sc[0][i];
#pragma unroll
        for (int c = 1; c < NTT; ++c) m = fmaxf(m, sc[c][i]);
        m = fmaxf(m, dpp<0xB1>(m)); m = fmaxf(m, dpp<0x4E>(m)); m = fmaxf(m, dpp<0x141>(m)); m = fmaxf(m, dpp<0x140>(m));
        const float ms = m == -INFINITY ? 0.f : m;           // a wave without live steps for this beam: every term exp2(-inf) = 0
        float lsum = 0.f;
#pragma unroll
        for (int c = 0; c < NTT; ++c) { sc[c][i] = __builtin_amdgcn_exp2f(sc[c][i] - ms); lsum += sc[c][i]; }
        lsum = row16_sum(lsum);
        mrow_[i] = m;
        if (l16 == 0) { ml[wv * WB + kq + 4 * i] = m; ml[(8 + wv) * WB + kq + 4 * i] = lsum; }
      }
      RV_STAMP(d, step, 14);
      __syncthreads();
      RV_STAMP(d, step, 15);
      {
        // merge: lane (beam = lane / 8, wave g = lane % 8) takes one (max, sum) pair; 8-lane butterflies give every lane of the
        // group the beam's maximum and its total (the same instruction sequence in every wave: the same bits in every wave)
        const float mv = ml[(lane & 7) * WB + (lane >> 3)], lv = ml[(8 + (lane & 7)) * WB + (lane >> 3)];
        float Mgl = mv;
        Mgl = fmaxf(Mgl, dpp<0xB1>(Mgl)); Mgl = fmaxf(Mgl, dpp<0x4E>(Mgl)); Mgl = fmaxf(Mgl, dpp<0x141>(Mgl));
        float tl = mv == -INFINITY ? 0.f : lv * __builtin_amdgcn_exp2f(mv - Mgl);
        tl += dpp<0xB1>(tl); tl += dpp<0x4E>(tl); tl += dpp<0x141>(tl);
        // all-masked chunk: 0 / 0 = NaN like the reference
        const float rt = Mgl == -INFINITY ? __int_as_float(0x7fc00000) : __builtin_amdgcn_rcpf(tl);   // (v_rcp_f32: 1 ulp; the IEEE division is a dozen dependent instructions on this phase's critical path)
        // alignments -> A fragments of the context product: [k-block t / 8][row mx_row(beam) (+ 1: low part)][t % 8] f16 of alpha 2^14 (in `part`,
        // idle between the gates and the cell product at the end of the step)
        _Float16* aa = reinterpret_cast<_Float16*>(part);
        float Mg_[NI], rr_[NI];                                // (both beams' pairs fetched before either is used: one LDS round trip, not two)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          Mg_[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(32 * (kq + 4 * i), __float_as_int(Mgl)));
          rr_[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(32 * (kq + 4 * i), __float_as_int(rt)));
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int beam = kq + 4 * i;                         // its merged pair sits in lanes 8 beam .. 8 beam + 7
          const float Mg = Mg_[i], rr = rr_[i];
          const bool nanrow = rr != rr && beam < W;
          const float f = (beam >= W || mrow_[i] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mrow_[i] - Mg) * rr * 16384.f;
#pragma unroll
          for (int c = 0; c < NTT; ++c) {
            const int t = 16 * (wv + 8 * c) + l16;
            if (t < 32 * NIT) {
              const float av = nanrow ? rr : sc[c][i] * f;
              const _Float16 hi = (_Float16)av, lo = (_Float16)(av - (float)hi);
              _Float16* q = aa + ((t >> 3) * 16 + mx_row(beam)) * 8 + (t & 7);
              q[0] = hi; q[8] = lo;
            }
          }
        }
      }
      __syncthreads();
      RV_STAMP(d, step, 5);
    } else {
    // ================= scores from the resident key rows: lane w keeps beam w's score of the even row of a pair, lane 8 + w
    //   that of the odd row
    float sc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) sc[p] = -INFINITY;
    if constexpr (BAH && MXC) {
      // score_t log2(e) = const + sum_j v'_j / (1 + 2^(k'_tj + q'_j)) (see the resident rows above): MODE 1 = E_k . E_q + 1, one reciprocal;
      // MODE 0 = the chunk kept k' (a key out of range); MODE 2 = this step has a query out of range
      auto bah_scores = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const float* qsrc = MODE == 1 ? fold : pqs;
#pragma unroll
        for (int w = 0; w < W; ++w) {
          float4 qv[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) qv[m] = *reinterpret_cast<const float4*>(&qsrc[w * RV_U + 16 * s8 + 4 * m]);
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            f2 pp = f2{0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const float4 vv = *reinterpret_cast<const float4*>(&vat[16 * s8 + 4 * m]);
              auto el = [&](float k, float q) -> float {
                if constexpr (MODE == 1) return __builtin_amdgcn_rcpf(fmaf(k, q, 1.0f));
                else if constexpr (MODE == 0) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(k + q));
                else return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(k) + q));
              };
              const float r0 = el(kr[p][m].x, qv[m].x), r1 = el(kr[p][m].y, qv[m].y), r2 = el(kr[p][m].z, qv[m].z), r3 = el(kr[p][m].w, qv[m].w);
              pp = __builtin_elementwise_fma(f2{vv.x, vv.y}, f2{r0, r1}, pp);
              pp = __builtin_elementwise_fma(f2{vv.z, vv.w}, f2{r2, r3}, pp);
            }
            float sw = pp.x + pp.y;
            sw += dpp<0xB1>(sw); sw += dpp<0x4E>(sw); sw += dpp<0x141>(sw);      // the 8 lanes of this half row
            const int it = 2 * p + half;
            sc[p] = (s8 == w && it < NIT && ((livebits >> it) & 1u)) ? sw : sc[p];
          }
        }
      };
      if (kbig) bah_scores(std::integral_constant<int, 0>{});
      else if (s_qbig) bah_scores(std::integral_constant<int, 2>{});
      else bah_scores(std::integral_constant<int, 1>{});
    } else
#pragma unroll
    for (int w = 0; w < W; ++w) {
      float4 qv[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) qv[m] = *reinterpret_cast<const float4*>(&(BAH ? pqs : qp)[w * RV_U + 16 * s8 + 4 * m]);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        f2 pp = f2{0.f, 0.f};
        if (BAH) {
          // Bahdanau, normalize = False: score_t = sum_j v_j tanh(keys_tj + pq_j); keys and pq carry 2 log2(e),
          // v carries -2 log2(e): score * log2(e) = const + sum_j v'_j / (1 + exp2(k'_tj + pq'_j)), the constant cancels in the softmax.
          // Raw v_exp_f32 (no denormal-range rescue: 1 + e does not see it; +inf gives rcp = 0 = tanh's limit)
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const float4 vv = *reinterpret_cast<const float4*>(&vat[16 * s8 + 4 * m]);
            const float r0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(kr[p][m].x + qv[m].x));
            const float r1 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(kr[p][m].y + qv[m].y));
            const float r2 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(kr[p][m].z + qv[m].z));
            const float r3 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(kr[p][m].w + qv[m].w));
            pp = __builtin_elementwise_fma(f2{vv.x, vv.y}, f2{r0, r1}, pp);
            pp = __builtin_elementwise_fma(f2{vv.z, vv.w}, f2{r2, r3}, pp);
          }
        } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          pp = __builtin_elementwise_fma(f2{kr[p][m].x, kr[p][m].y}, f2{qv[m].x, qv[m].y}, pp);
          pp = __builtin_elementwise_fma(f2{kr[p][m].z, kr[p][m].w}, f2{qv[m].z, qv[m].w}, pp);
        }
        }
        float sw = pp.x + pp.y;
        sw += dpp<0xB1>(sw);    // quad_perm [1,0,3,2]
        sw += dpp<0x4E>(sw);    // quad_perm [2,3,0,1]
        sw += dpp<0x141>(sw);   // row_half_mirror: the other quad of this 8-lane half
        const int it = 2 * p + half;
        sc[p] = (s8 == w && it < NIT && ((livebits >> it) & 1u)) ? sw : sc[p];
      }
    }
    RV_STAMP(d, step, 4);
    // ================= softmax over the chunk's T_m steps: per-stream (max, sum), one fixed-order merge per beam
    {
      float m = -INFINITY;
#pragma unroll
      for (int p = 0; p < NP; ++p) m = fmaxf(m, sc[p]);
      m = fmaxf(m, dpp<0x128>(m));                          // row_ror:8 -- the stream's other half
      float lsum = 0.f;
#pragma unroll
      for (int p = 0; p < NP; ++p) { sc[p] = exp2f(sc[p] - m); lsum += sc[p]; }   // stream without live rows: NaN, dropped below
      lsum += dpp<0x128>(lsum);
      if (sub < W) { ml[sid * WB + sub] = m; ml[(32 + sid) * WB + sub] = m == -INFINITY ? 0.f : lsum; }
      __syncthreads();
      if (wv < W) {
        const float ms = lane < 32 ? ml[lane * WB + wv] : -INFINITY;
        const float ls = lane < 32 ? ml[(32 + lane) * WB + wv] : 0.f;
        const float Mg = wave_max_fast(ms);
        const float tot = wave_sum_fast(ms == -INFINITY ? 0.f : ls * exp2f(ms - Mg));   // all-masked chunk: 0 -> NaN like the reference
        if (lane == 0) { mg[wv] = Mg; mg[WB + wv] = Mg == -INFINITY ? __int_as_float(0x7fc00000) : 1.0f / tot; }
      }
      __syncthreads();
      const float f = (s8 < W && m != -INFINITY) ? exp2f(m - mg[s8]) * mg[WB + s8] : 0.f;
      const float nanv = (s8 < W && mg[WB + s8] != mg[WB + s8]) ? mg[WB + s8] : 0.f;
#pragma unroll
      for (int p = 0; p < NP; ++p) sc[p] = m == -INFINITY ? nanv : sc[p] * f;   // alignments of beam s8 on this half's rows
    }
    RV_STAMP(d, step, 5);
    if constexpr (MX) {
      // ================= (Bahdanau on the matrix pipe) the alignments this lane holds -- beam s8 on rows t = sid + 32 (2 p + half) -- as the
      //   context product's A fragments: [k-block t / 8][row mx_row(beam) (+ 1: low part)][t % 8] f16 of alpha 2^14, in `part` (idle
      //   between the gates and the cell product at the end of the step); padded steps carry an alignment of exactly 0
      if (s8 < W) {
        _Float16* aa = reinterpret_cast<_Float16*>(part);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int it = 2 * p + half, t = sid + 32 * it;
          if (it < NIT) {
            const float av = sc[p] * 16384.f;
            const _Float16 hi = (_Float16)av, lo = (_Float16)(av - (float)hi);
            _Float16* q = aa + ((t >> 3) * 16 + mx_row(s8)) * 8 + (t & 7);
            q[0] = hi; q[8] = lo;
          }
        }
      }
      __syncthreads();
    } else {
    // ================= attention-layer context part = sum_t alpha_t * U'_t, one beam at a time (8-register accumulator)
    {
      float4* slab = reinterpret_cast<float4*>(fold + (size_t)wv * (4 * 2 * 16 * 4));
#pragma unroll
      for (int w = 0; w < W; ++w) {
        f2 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = f2{0.f, 0.f};
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const float al = row_bcast16(sc[it >> 1], 8 * (it & 1) + w);
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            a[2 * m] = __builtin_elementwise_fma(f2{al, al}, f2{ur[it][m].x, ur[it][m].y}, a[2 * m]);
            a[2 * m + 1] = __builtin_elementwise_fma(f2{al, al}, f2{ur[it][m].z, ur[it][m].w}, a[2 * m + 1]);
          }
        }
        // fold the wave's 4 streams (same columns) in ONE LDS round trip: every row parks its 2 column blocks, rows 0 and 1
        // then sum one block each over the 4 streams in fixed order.  DS operations of one wave execute in issue order, so
        // the reads see the writes without a wait in between.
#pragma unroll
        for (int m = 0; m < 2; ++m) slab[(row * 2 + m) * 16 + sub] = make_float4(a[2 * m].x, a[2 * m].y, a[2 * m + 1].x, a[2 * m + 1].y);
        asm volatile("" ::: "memory");
        if (row < 2) {
          const float4 p0 = slab[(0 * 2 + row) * 16 + sub], p1 = slab[(1 * 2 + row) * 16 + sub];
          const float4 p2 = slab[(2 * 2 + row) * 16 + sub], p3 = slab[(3 * 2 + row) * 16 + sub];
          *reinterpret_cast<float4*>(&ctxp[(wv * W + w) * RV_U + 4 * sub + 64 * row]) =
              make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y, ((p0.z + p1.z) + p2.z) + p3.z, ((p0.w + p1.w) + p2.w) + p3.w);
        }
        asm volatile("" ::: "memory");
      }
    }
    __syncthreads();
    RV_STAMP(d, step, 6);
    for (int i = tid; i < W * RV_U; i += NT) {              // ctx' (8 waves, fixed order) [+ h . A_h (16 K groups) = attention vector when D == 2]
      const int w = i >> 7, col = i & 127;
      float s0 = 0.f;
      if (D > 1) {
#pragma unroll
        for (int g = 0; g < 16; ++g) s0 += part[(g * W + w) * RV_U + col];
      }
      float s1 = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) s1 += ctxp[(g * W + w) * RV_U + col];
      const float av = s0 + s1;
      att[i] = av; attT[col * WB + w] = av;
    }
    __syncthreads();
    
    }
    }
    if constexpr (MX) {
      const int l16 = lane & 15, kq = lane >> 4;
      constexpr int NI = W > 4 ? 2 : 1;
      // ================= attention-layer context part = sum_t alpha_t U'_t on the matrix pipe: rows = beams, this wave's 16 units,
      //   K = the chunk's steps; the product is complete in one wave (no partial sums to merge)
      {
        const _Float16* aa = reinterpret_cast<const _Float16*>(part) + (kq * 16 + l16) * 8;
        f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;           // against the high / the low part of U'
#pragma unroll
        for (int ks = 0; ks < NIT; ++ks) {
          const h8 a = *reinterpret_cast<const h8*>(aa + ks * 512);
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, ub[ks][0]), a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, ub[ks][1]), a1, 0, 0, 0);
        }
        float acc[2];                                      // beams kq, kq + 4
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = (a0[2 * i] + a0[2 * i + 1]) + (a1[2 * i] + a1[2 * i + 1]);
        RV_STAMP(d, step, 6);
        const int col = 16 * wv + l16;
#pragma unroll
        for (int i = 0; i < NI; ++i)
          if (kq + 4 * i < W) {
            const float av = acc[i] * d.mx_udescale;
            if constexpr (!MXC) att[(kq + 4 * i) * RV_U + col] = av;
            if constexpr (MXC) {   // ctx' as A fragments of the cell product: k = col, ctx' . mx_uscale (= acc 2^-14, below 2^14) in two f16 parts
              const float sv = acc[i] * (1.0f / 16384.f);
              const _Float16 hi = (_Float16)sv, lo = (_Float16)(sv - (float)hi);
              _Float16* xq = xim + ((col >> 3) * 16 + mx_row(kq + 4 * i)) * 8 + (col & 7);
              xq[0] = hi; xq[8] = lo;
            } else attT[col * WB + kq + 4 * i] = av;
          }
      }
      __syncthreads();
    }
    RV_STAMP(d, step, 8);
    // ================= logits = attention . W_fc + b_fc
    if constexpr (MXC) {
      // Matrix-pipe cell product: the logits are ONE 16-column tile over the beams' [ctx' | h] image (the cell product's A operand),
      // 24 MFMAs that wave 0 takes alone, B fragments from the LDS copy of the 16 KB Wl16 image (from L2 the same reads queue behind
      // the other waves' weight stream: 4.4 k cycles) -- the other waves go straight on to the cell product (they only need ctx' and
      // h): no logits phase and no barrier in front of the beam step.
      if (wv == 0) {
        const int l16 = lane & 15, kq = lane >> 4;
        const _Float16* xa = xim + (kq * 16 + l16) * 8;
        const uint4* wl = reinterpret_cast<const uint4*>(dsm + L.wl16) + lane;
        f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          uint4 bh[4], bl[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) { bh[i] = wl[(2 * (4 * half + i)) * 64]; bl[i] = wl[(2 * (4 * half + i) + 1) * 64]; }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const h8 a = *reinterpret_cast<const h8*>(xa + (4 * half + i) * 512);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, bh[i]), a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, bl[i]), a1, 0, 0, 0);
          }
        }
        float acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = (a0[2 * i] + a0[2 * i + 1]) + (a1[2 * i] + a1[2 * i + 1]);
        constexpr int NI = W > 4 ? 2 : 1;               // C/D: lane (column v = l16, kq) holds beams kq + 4 i
#pragma unroll
        for (int i = 0; i < NI; ++i)
          if (kq + 4 * i < W && l16 < V) lg[(kq + 4 * i) * RV_MAX_VOCAB + l16] = acc[i] * d.mx_ldescale + s_wfc[BFC + l16];
      }
    } else {
    {
      const int o8 = tid >> 3, s8 = tid & 7;
      if (o8 < W * V) {                                     // whole 8-lane groups take the branch together
        const int w = o8 / V, v = o8 % V;
        // lane s8 takes k = 16 s8 .. 16 s8 + 15: four float4 of the attention vector against four of the weight row
        f2 pp = f2{0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float4 a4 = *reinterpret_cast<const float4*>(&att[w * RV_U + 16 * s8 + 4 * m]);
          const float4 w4 = *reinterpret_cast<const float4*>(&s_wfc[v * FCW + 16 * s8 + 4 * m]);
          pp = __builtin_elementwise_fma(f2{a4.x, a4.y}, f2{w4.x, w4.y}, pp);
          pp = __builtin_elementwise_fma(f2{a4.z, a4.w}, f2{w4.z, w4.w}, pp);
        }
        if (D == 1) {                                       // + h . (A_h W_fc)
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const float4 a4 = *reinterpret_cast<const float4*>(&qp[w * RV_U + 16 * s8 + 4 * m]);
            const float4 w4 = *reinterpret_cast<const float4*>(&s_nh[v * FCW + 16 * s8 + 4 * m]);
            pp = __builtin_elementwise_fma(f2{a4.x, a4.y}, f2{w4.x, w4.y}, pp);
            pp = __builtin_elementwise_fma(f2{a4.z, a4.w}, f2{w4.z, w4.w}, pp);
          }
        }
        float p = pp.x + pp.y;
        p += dpp<0xB1>(p);    // quad_perm [1,0,3,2]
        p += dpp<0x4E>(p);    // quad_perm [2,3,0,1]
        p += dpp<0x141>(p);   // row_half_mirror: the other quad of this 8-lane group
        if (s8 == 0) lg[w * RV_MAX_VOCAB + v] = p + s_wfc[BFC + v];
      }
    }
    __syncthreads();
    }
    RV_STAMP(d, step, 9);
    // ================= beam step (wave 0): log-softmax, finished masking, top-W, bookkeeping
    if (tid < 64) {
      const int w = lane / V, v = lane % V;
      const bool cand = lane < W * V;
      const size_t o = ((size_t)step * d.B + b) * W;
      float val = -INFINITY;
      if (d.greedy) {                  // GreedyEmbeddingSampler: argmax of the raw logits (W == 1)
        if (cand) { val = lg[v]; d.step_logits[((size_t)step * d.B + b) * V + v] = val; }
      } else if (cand) {
        if (d.step_logits) d.step_logits[(((size_t)step * d.B + b) * W + w) * V + v] = lg[w * RV_MAX_VOCAB + v];   // debug tap (option persist_taps)
        float lv[RV_MAX_VOCAB];                             // the beam's logits in two LDS reads; same order of operations as the loops
        *reinterpret_cast<float4*>(lv) = *reinterpret_cast<const float4*>(&lg[w * RV_MAX_VOCAB]);
        *reinterpret_cast<float4*>(lv + 4) = *reinterpret_cast<const float4*>(&lg[w * RV_MAX_VOCAB + 4]);
        float m = lv[0];
#pragma unroll
        for (int x = 1; x < RV_MAX_VOCAB; ++x) m = x < V ? fmaxf(m, lv[x]) : m;
        float ssum = 0.f, mine = lv[0];
#pragma unroll
        for (int x = 0; x < RV_MAX_VOCAB; ++x) { ssum += x < V ? __expf(lv[x] - m) : 0.f; mine = x == v ? lv[x] : mine; }
        const float lse = __logf(ssum);
        const bool fin = s_fin[w] != 0;
        const float lp = fin ? (v == d.end_token ? 0.f : -FLT_MAX) : (mine - m) - lse;
        val = s_lprob[w] + lp;
      }
      bool taken = !cand;
      int my_word = 0, my_par = 0; float my_val = 0.f;
      for (int k = 0; k < W; ++k) {
        const float mx = wave_max_fast(taken ? -INFINITY : val);
        const unsigned long long hit = __ballot(!taken && (val == mx || mx == -INFINITY));
        const int win = __ffsll((long long)hit) - 1;
        const float wval = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(val), win));
        if (lane == win) taken = true;
        if (lane == k) { my_word = win % V; my_par = win / V; my_val = wval; }
      }
      bool nf = false; int nl = 0;
      if (lane < W) {
        const bool pf = s_fin[my_par] != 0;
        nf = pf || my_word == d.end_token;
        nl = s_len[my_par] + (pf ? 0 : 1);
      }
      const unsigned long long fmask = __ballot(lane < W && nf);
      if (d.greedy) {
        // BasicDecoder without impute_finished: a finished row keeps sampling and the loop runs until EVERY row of the
        // slab has finished, so a chunk may only stop once all chunks have reported their first finished step
        // (nfin[0] = how many have, nfin[1] = the latest of those steps + 1) and it has itself recorded that many steps.
        if (lane == 0) {
          const bool pf = s_fin[0] != 0;
          d.step_ids[(size_t)step * d.B + b] = my_word;
          s_tok[0] = my_word; s_fin[0] = nf; s_parent[0] = 0;
          if (nf && !pf) { s_len[0] = step + 1; atomicMax(&d.nfin[1], step + 1); __threadfence(); atomicAdd(&d.nfin[0], 1); }
          // writer: atomicMax(step) -> fence -> atomicAdd(count); reader: count, ACQUIRE fence, then the step, so a reader that
          // sees every chunk counted also sees every chunk's first-finish step (the two words sit in different L2 channels).
          // The acquire fence is only paid once the count is complete (the last few steps of the slab).
          const int cnt = __hip_atomic_load(&d.nfin[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          int stop = 0;
          if (nf && cnt >= d.B) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int sg = __hip_atomic_load(&d.nfin[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            stop = step + 1 >= sg;
          }
          s_allfin = stop;
        }
      } else {
      if (lane < W) {      // all reads of the old bookkeeping happened above (same wave, program order)
        d.step_ids[o + lane] = my_word; d.parent_ids[o + lane] = my_par; d.step_scores[o + lane] = my_val;
        s_tok[lane] = my_word; s_fin[lane] = nf; s_lprob[lane] = my_val; s_len[lane] = nl; s_parent[lane] = my_par;
      }
      if (lane == 0) s_allfin = __popcll(fmask) == W;
      }
      RV_STAMP(d, step, 11);      // end of the beam step (wave 0)
    }
    // ================= next step's cell product on THIS step's beams: z~[w] = [attention_w | h_w] . Wcat (the beam step above
    //   only decides which z~ each new beam inherits).  thread = (4 gate columns c4, K group kg); waves 1-7 start at once,
    //   wave 0 joins after the beam step, so K group 0 (waves 0-1) is the short one: rows 0-39 | 40-111 | 112-183 | 184-255.
    if constexpr (MXC2) {
      // Two cells on the matrix pipe.  End of the step, on THIS step's beams (the beam step only decides which rows each new beam inherits):
      //   cell 0's next product  z~_0 = [ctx' | h_1 | h_0] . [W_a ; A_h W_a ; U_0]   (K = 384: jobs 0..47, pair j = 4 ks + g of the Wc16 image)
      //   cell 1's recurrent one z~_1 = h_1 . U_1                                     (K = 128: jobs 48..63, pairs 16..31 of the W1c16 image)
      // -- the attention vector never exists: its h_1 . A_h half is folded into cell 0's rows 128..255 and into the output layer at
      // load time, its context half is the ctx' segment.  Wave wv owns units 16 wv .. 16 wv + 15 of all four gates of both products
      // (64 jobs of 2 KB); wave 0 -- output layer and beam step first -- finds its first MXC2_CACHE jobs in LDS.
      if (step + 1 < steps) {
        RV_STAMP_W1(d, step, 12);
        const int l16 = lane & 15, kq = lane >> 4;
        const uint4* w0img = reinterpret_cast<const uint4*>(d.Wc16) + (size_t)wv * (48 * 128) + lane;
        const uint4* w1img = reinterpret_cast<const uint4*>(d.W1c16) + (size_t)wv * (32 * 128) + 16 * 128 + lane;
        const _Float16* xa = xim + (kq * 16 + l16) * 8;
        constexpr int NI = W > 4 ? 2 : 1, NB = 4;
        f4v acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f4v{0.f, 0.f, 0.f, 0.f};
        auto src = [&](int j, int q) -> uint4 { return j < 48 ? w0img[(2 * j + q) * 64] : w1img[(2 * (j - 48) + q) * 64]; };
        auto mm = [&](int j, const uint4& vh, const uint4& vl) {
          const int ks = j < 48 ? (j >> 2) : 4 + ((j - 48) >> 2), g = j & 3;      // h_1 . U_1 reads the h_1 segment: k-steps 4..7
          const h8 a = *reinterpret_cast<const h8*>(xa + ks * 512);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, vl), acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, vh), acc[g], 0, 0, 0);
        };
        auto flush = [&](float* dst, float scale) {         // C/D: lane (column l16, kq): registers 0 + 1 = beam kq, 2 + 3 = beam kq + 4
#pragma unroll
          for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
              if (kq + 4 * i < W) dst[(kq + 4 * i) * ZS + RV_U * g + 16 * wv + l16] = (acc[g][2 * i] + acc[g][2 * i + 1]) * scale;
            acc[g] = f4v{0.f, 0.f, 0.f, 0.f};
          }
        };
        auto stream = [&](auto first_tag) {
          constexpr int J0 = decltype(first_tag)::value;    // jobs J0..63 come from L2 through a window of NB
          uint4 bh[NB], bl[NB];
#pragma unroll
          for (int i = 0; i < NB; ++i) { bh[i] = src(J0 + i, 0); bl[i] = src(J0 + i, 1); }
          if constexpr (J0 > 0) {                            // (wave 0) the LDS-resident jobs first: the first requests are in flight meanwhile
            const uint4* wc = reinterpret_cast<const uint4*>(wcache) + lane;
            uint4 fh = wc[0], fl = wc[64];
#pragma unroll
            for (int j = 0; j < J0; ++j) {
              uint4 nh = fh, nl = fl;
              if (j + 1 < J0) { nh = wc[(2 * (j + 1)) * 64]; nl = wc[(2 * (j + 1) + 1) * 64]; }
              mm(j, fh, fl);
              fh = nh; fl = nl;
              __builtin_amdgcn_sched_barrier(0);
            }
          }
#pragma unroll
          for (int j = J0; j < 64; ++j) {
            mm(j, bh[(j - J0) % NB], bl[(j - J0) % NB]);
            __builtin_amdgcn_sched_barrier(0);
            if (j + NB < 64) { bh[(j - J0) % NB] = src(j + NB, 0); bl[(j - J0) % NB] = src(j + NB, 1); }
            if (j == 47) flush(part, d.mx_cdescale);
            __builtin_amdgcn_sched_barrier(0);
          }
          flush(partU, d.mx_c1descale);
        };
        if (wv == 0) stream(std::integral_constant<int, MXC2_CACHE>{});
        else stream(std::integral_constant<int, 0>{});
        RV_STAMP_W1(d, step, 13);
      }
    } else if constexpr (MXC) {
      // Matrix pipe: z~ [beams x 512] = x [beams x 256] . Wcat2 as split-f16 MFMAs.  Wave wv owns units 16 wv .. 16 wv + 15 of all
      // four gates over the whole K (no partial sums to merge): 32 (k-step, gate) pairs, each one B fragment pair (high, low
      // part; 2 KB per wave) streamed from the Wc16 image through a rolling window of NB pairs, the last NC pairs from LDS.  A = the
      // beams' [ctx' | h] image (rows = beam slots as in the scores), three exact part products per pair.
      if (step + 1 < steps) {
        RV_STAMP_W1(d, step, 12);
        const int l16 = lane & 15, kq = lane >> 4;
        const uint4* wimg = reinterpret_cast<const uint4*>(d.Wc16) + (size_t)wv * (32 * 128) + lane;   // pair p, part q: [(2 p + q) * 64]
        const _Float16* xa = xim + (kq * 16 + l16) * 8;      // k-step ks: + 512 ks (both parts of a beam in rows mx_row, + 1)
        constexpr int NS = 32 - NC - NR, NB = 4;             // streamed pairs
        f4v acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f4v{0.f, 0.f, 0.f, 0.f};
        auto mm = [&](int p, const uint4& vh, const uint4& vl) {
          const int ks = p >> 2, g = p & 3;
          const h8 a = *reinterpret_cast<const h8*>(xa + ks * 512);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, vl), acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, vh), acc[g], 0, 0, 0);
        };
        // Wave 0 takes the output layer and the beam step first (4.7 k cycles); its columns' 32 pairs all sit in LDS (64 KB) and it takes
        // them afterwards, half a k-step (two gates) at a time with the next half's fragments requested before this half's MFMAs.  (As a
        // chain of whole k-steps, each waiting for its eight LDS reads -- round 3 -- the 32 pairs took 4.3 k cycles and the step ended
        // 1.5 k cycles after the streaming waves had finished: C3 launch 0.342 -> 0.325 ms, R 0.916 -> 0.842 on one box, round-robin.
        // Handing wave 0's columns to waves 1-4, one gate each between their own streamed pairs, measured SLOWER than either, 0.35 / 0.88:
        // the extra LDS round trips delay the streaming waves' next requests, and with T_m > 256 there is no register for a prefetch.)
        if (wv == 0) {
          const uint4* wc = reinterpret_cast<const uint4*>(wcache) + lane;       // pair p = 4 ks + g, part q: [(2 p + q) * 64]
          uint4 fr[2][4];                                    // [buffer][(g0 high, g0 low, g1 high, g1 low)]
#pragma unroll
          for (int j = 0; j < 4; ++j) fr[0][j] = wc[j * 64];
#pragma unroll
          for (int hs = 0; hs < 16; ++hs) {                  // half k-steps: k-step hs / 2, gates 2 (hs % 2), + 1
            if (hs + 1 < 16) {
#pragma unroll
              for (int j = 0; j < 4; ++j) fr[(hs + 1) & 1][j] = wc[(4 * (hs + 1) + j) * 64];
            }
            const h8 a = *reinterpret_cast<const h8*>(xa + (hs >> 1) * 512);
            const int g0 = 2 * (hs & 1);
            acc[g0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, fr[hs & 1][1]), acc[g0], 0, 0, 0);
            acc[g0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, fr[hs & 1][3]), acc[g0 + 1], 0, 0, 0);
            acc[g0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, fr[hs & 1][0]), acc[g0], 0, 0, 0);
            acc[g0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, fr[hs & 1][2]), acc[g0 + 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          uint4 bh[NB], bl[NB];
#pragma unroll
          for (int i = 0; i < NB; ++i) { bh[i] = wimg[(2 * i) * 64]; bl[i] = wimg[(2 * i + 1) * 64]; }
          {                                                  // the on-chip pairs first: the first requests are in flight meanwhile
            const uint4* wc = reinterpret_cast<const uint4*>(wcache) + 32 * 128 + (size_t)(wv - 1) * (NC * 128) + lane;
#pragma unroll
            for (int c = 0; c < NC; ++c) mm(32 - NC + c, wc[(2 * c) * 64], wc[(2 * c + 1) * 64]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int p = 0; p < NS; ++p) {
            mm(p, bh[p % NB], bl[p % NB]);
            __builtin_amdgcn_sched_barrier(0);
            if (p + NB < NS) { bh[p % NB] = wimg[(2 * (p + NB)) * 64]; bl[p % NB] = wimg[(2 * (p + NB) + 1) * 64]; }
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (NR > 0) {                            // the pairs that live in registers
#pragma unroll
            for (int r = 0; r < NR; ++r) mm(NS + r, rbh[r], rbl[r]);
          }
        }
        // C/D: lane (column l16, kq) holds rows 4 kq + i: registers 0 + 1 = beam kq (high- and low-part rows), 2 + 3 = beam kq + 4
        constexpr int NI = W > 4 ? 2 : 1;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int i = 0; i < NI; ++i)
            if (kq + 4 * i < W) part[(kq + 4 * i) * ZS + RV_U * g + 16 * wv + l16] = (acc[g][2 * i] + acc[g][2 * i + 1]) * d.mx_cdescale;
        RV_STAMP_W1(d, step, 13);
      }
    } else
    if (step + 1 < steps) {
      const int c4 = tid & 127, kg = tid >> 7;
      // one cell: rows 0-39 | 40-111 | 112-183 | 184-255;  two cells: 64 rows each here, and K groups 1-3 also take the
      // second cell's recurrent product below (48 / 40 / 40 rows)
      const int kb = D > 1 ? 64 * kg : (kg == 0 ? 0 : 72 * kg - 32), ke = D > 1 ? kb + 64 : (kg == 0 ? 40 : kb + 72);
      f2 acc[W][2];
#pragma unroll
      for (int w = 0; w < W; ++w) { acc[w][0] = f2{0.f, 0.f}; acc[w][1] = f2{0.f, 0.f}; }
      const float* wc = Wcat + 4 * c4;
      auto fma_row = [&](const float4& wv, const float* xrow) {
        float xv[WB];
        *reinterpret_cast<float4*>(xv) = *reinterpret_cast<const float4*>(xrow);
        if (W > 4) *reinterpret_cast<float4*>(xv + 4) = *reinterpret_cast<const float4*>(xrow + 4);
#pragma unroll
        for (int w = 0; w < W; ++w) {
          acc[w][0] = __builtin_elementwise_fma(f2{xv[w], xv[w]}, f2{wv.x, wv.y}, acc[w][0]);
          acc[w][1] = __builtin_elementwise_fma(f2{xv[w], xv[w]}, f2{wv.z, wv.w}, acc[w][1]);
        }
      };
      const bool cached = CACHE && kg > 0;                  // wave-uniform: K groups 1-3 find their last 8 rows in LDS
      const int ke_g = cached ? ke - 8 : ke;                // rows streamed from L2
      if (cached) {                                         // the rows that are already on chip
        const int kc = ke - 8;
        const float* xk = kc < RV_U ? attT + kc * WB : hcT + (kc - RV_U) * WB;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          fma_row(*reinterpret_cast<const float4*>(&wcache[((kg - 1) * 8 + i) * RV_G + 4 * c4]), xk + i * WB);
      }
#pragma unroll 1
      for (int k0 = kb; k0 < ke_g; k0 += 8) {               // batches never straddle row 128 (40, 112 and 184 are multiples of 8)
        float4 wr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) wr[i] = *reinterpret_cast<const float4*>(wc + (size_t)(k0 + i) * RV_G);
        __builtin_amdgcn_sched_barrier(0);
        const float* xk = k0 < RV_U ? attT + k0 * WB : (D > 1 ? h0T : hcT) + (k0 - RV_U) * WB;
#pragma unroll
        for (int i = 0; i < 8; ++i) fma_row(wr[i], xk + i * WB);
      }
#pragma unroll
      for (int w = 0; w < W; ++w)
        *reinterpret_cast<float4*>(&part[(kg * W + w) * RV_G + 4 * c4]) = make_float4(acc[w][0].x, acc[w][0].y, acc[w][1].x, acc[w][1].y);
      if (D > 1 && kg > 0) {            // h_1 . U_1 of this step's beams (U_1 = rows 128..255 of Wcat1; h_1 = hcT's h rows)
        const int rb = kg == 1 ? 0 : 8 + 40 * (kg - 1), re = kg == 1 ? 48 : rb + 40;
#pragma unroll
        for (int w = 0; w < W; ++w) { acc[w][0] = f2{0.f, 0.f}; acc[w][1] = f2{0.f, 0.f}; }
        const float* wu = Wcat1 + (size_t)RV_U * RV_G + 4 * c4;
#pragma unroll 1
        for (int k0 = rb; k0 < re; k0 += 8) {
          float4 wr[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) wr[i] = *reinterpret_cast<const float4*>(wu + (size_t)(k0 + i) * RV_G);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float xv[WB];
            *reinterpret_cast<float4*>(xv) = *reinterpret_cast<const float4*>(&hcT[(k0 + i) * WB]);
            if (W > 4) *reinterpret_cast<float4*>(xv + 4) = *reinterpret_cast<const float4*>(&hcT[(k0 + i) * WB + 4]);
#pragma unroll
            for (int w = 0; w < W; ++w) {
              acc[w][0] = __builtin_elementwise_fma(f2{xv[w], xv[w]}, f2{wr[i].x, wr[i].y}, acc[w][0]);
              acc[w][1] = __builtin_elementwise_fma(f2{xv[w], xv[w]}, f2{wr[i].z, wr[i].w}, acc[w][1]);
            }
          }
        }
#pragma unroll
        for (int w = 0; w < W; ++w)
          *reinterpret_cast<float4*>(&partU[((kg - 1) * W + w) * RV_G + 4 * c4]) = make_float4(acc[w][0].x, acc[w][0].y, acc[w][1].x, acc[w][1].y);
      }
    }
    __syncthreads();
    RV_STAMP(d, step, 10);
    if (s_allfin) { done_steps = step + 1; break; }         // uniform: every thread reads the same LDS word
  }
  if (tid < W) { d.lengths[row0 + tid] = s_len[tid]; d.finished[row0 + tid] = (uint8_t)s_fin[tid]; }
  if (tid == 0) d.chunk_steps[b] = d.greedy ? (s_fin[0] ? s_len[0] : steps) : done_steps;   // greedy: the row's first finished step + 1
}

// One 64-thread workgroup per chunk: the chunk's [S,W] ids/parents are staged in LDS so the
// serial gather_tree back-trace runs on LDS latency, not on dependent global loads.
__global__ __launch_bounds__(64) void k_dec_finalize(DecState d, int32_t* tokens, float* out2, const void* const* ptab) {
  __shared__ int s_ids[64 * RV_MAX_BEAM], s_par[64 * RV_MAX_BEAM], s_tok[64];
  __shared__ int s_S;
  if (ptab) {      // graph replay: this call's output addresses (the kernel arguments are the capture's)
    tokens = static_cast<int32_t*>(const_cast<void*>(ptab[RV_PTAB_TOKENS]));
    out2 = static_cast<float*>(const_cast<void*>(ptab[RV_PTAB_OUT2]));
  }
  const int b = blockIdx.x, tid = threadIdx.x;
  const int steps = d.L - 1, W = d.W, V = d.V;
  // S: steps the reference loop runs for the WHOLE slab (until every row is finished); So: steps this
  // sub-slab actually ran.  For s in [So, S) all of its beams are finished: the reference emits the
  // end token at an unchanged top-1 score there, which is what is written below.
  // persistent decode: S of the slab = the slowest chunk's step count, taken here by every workgroup (B <= a few thousand ints
  // from L2) instead of in a launch of its own; workgroup 0 leaves it in S_dev for the host
  if (d.chunk_steps) {
    int m = 0;
    for (int i = tid; i < d.B; i += 64) m = max(m, d.chunk_steps[i]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    if (tid == 0) { s_S = m; if (b == 0) { d.S_dev[0] = m; d.S_dev[1] = m; d.S_host[0] = m; } }
    __syncthreads();
  }
  const int S = d.chunk_steps ? s_S : d.S_dev[0], So = d.chunk_steps ? d.chunk_steps[b] : d.S_dev[1 + d.part];
  int32_t* tk = tokens + (size_t)b * steps;
  if (d.greedy) {
    for (int s = tid; s < steps; s += 64) {
      tk[s] = s < S ? d.step_ids[(size_t)s * d.B + b] : d.pad_token;
      for (int v = 0; v < V; ++v)
        out2[((size_t)b * steps + s) * V + v] = s < S ? d.step_logits[((size_t)s * d.B + b) * V + v] : 0.f;
    }
    return;
  }
  for (int i = tid; i < So * W; i += 64) {
    const int s = i / W, w = i % W;
    s_ids[i] = d.step_ids[((size_t)s * d.B + b) * W + w];
    s_par[i] = d.parent_ids[((size_t)s * d.B + b) * W + w];
  }
  __syncthreads();
  if (tid == 0) {
    int maxlen = 0;
    for (int w = 0; w < W; ++w) maxlen = max(maxlen, d.lengths[(size_t)b * W + w]);
    const int Lb = min(S, maxlen);
    for (int s = 0; s < S; ++s) s_tok[s] = d.end_token;
    if (Lb > 0) {
      s_tok[Lb - 1] = s_ids[(Lb - 1) * W];
      int p = s_par[(Lb - 1) * W];
      for (int t = Lb - 2; t >= 0; --t) {
        s_tok[t] = s_ids[t * W + p];
        p = s_par[t * W + p];
      }
      bool done = false;
      for (int t = 0; t < Lb; ++t) {
        if (done) s_tok[t] = d.end_token;
        else if (s_tok[t] == d.end_token) done = true;
      }
    }
  }
  __syncthreads();
  // one wave, lane = step (max_output_len <= 64)
  const int s = tid;
  const int tokv = s < S ? s_tok[s] : d.pad_token;
  const float scv = s < S ? d.step_scores[((size_t)min(s, So - 1) * d.B + b) * W] : 0.f;
  if (s < steps) { tk[s] = tokv; out2[(size_t)b * steps + s] = scv; }
  if (d.call_bases) {
    // fused tokens_to_nuc_sequences + calc_prob_logits_beam_search_scores for this chunk
    const float prev = __shfl_up(scv, 1);
    const uint8_t ch = (s < S && tokv >= 0 && tokv < RV_MAX_VOCAB) ? d.lut[tokv] : 0;
    const unsigned long long m = __ballot(ch != 0);
    const int pos = __popcll(m & ((1ull << s) - 1ull));
    if (s < steps) {
      d.call_probs[(size_t)b * steps + s] = s < S ? __expf(scv - (s == 0 ? 0.f : prev)) : 0.f;
      d.call_bases[(size_t)b * steps + s] = 0;
    }
    __syncthreads();
    if (ch) d.call_bases[(size_t)b * steps + pos] = ch;
    if (s == 0) d.call_len[b] = __popcll(m);
  }
}

// one wave: lane s looks at step s (max_output_len <= 64), ballot finds the first finished step
__global__ __launch_bounds__(64) void k_dec_reduce_steps(DecParts p) {
  const int lane = threadIdx.x;
  int S = 0;
  for (int g = 0; g < p.n; ++g) {
    const bool done = lane < p.steps && p.nfin[g][lane] >= p.B[g];
    const unsigned long long m = __ballot(done);
    const int Sg = m ? __ffsll((long long)m) : p.steps;      // first finished step index + 1
    if (lane == 0) p.S_dev[1 + g] = Sg;
    S = max(S, Sg);
  }
  if (lane == 0) { p.S_dev[0] = S; p.S_host[0] = S; }
}

}  // namespace

template <int W, int D, int ATT>
static void launch_persist_wd(const DecState& d, const float* Wcat, const float* Wtok, const float* bdec,
                              const float* Wcat1, const float* bdec1, const float* Nh, hipStream_t s) {
  const size_t shm = sizeof(float) * PersistLds(W, D, ATT).total;
  if (d.Tm <= 64) hipLaunchKernelGGL((k_dec_persist<W, 2, D, ATT>), dim3(d.B), dim3(512), shm, s, d, Wcat, Wtok, bdec, Wcat1, bdec1, Nh);
  else if (d.Tm <= 256) hipLaunchKernelGGL((k_dec_persist<W, 8, D, ATT>), dim3(d.B), dim3(512), shm, s, d, Wcat, Wtok, bdec, Wcat1, bdec1, Nh);
  else hipLaunchKernelGGL((k_dec_persist<W, 11, D, ATT>), dim3(d.B), dim3(512), shm, s, d, Wcat, Wtok, bdec, Wcat1, bdec1, Nh);
}
template <int W>
static void launch_persist_w(const DecState& d, const float* Wcat, const float* Wtok, const float* bdec,
                             const float* Wcat1, const float* bdec1, const float* Nh, hipStream_t s) {
  if constexpr (W <= 5) {
    if (d.depth > 1 && d.mx_attention == 2) { launch_persist_wd<W, 2, 3>(d, Wcat, Wtok, bdec, Wcat1, bdec1, Nh, s); return; }   // two cells, everything on the matrix pipe
    if (d.depth > 1) { launch_persist_wd<W, 2, 0>(d, Wcat, Wtok, bdec, Wcat1, bdec1, Nh, s); return; }
  }
  if (d.attention == 1 && d.mx_attention == 2) launch_persist_wd<W, 1, 4>(d, Wcat, Wtok, bdec, Wcat1, bdec1, Nh, s);   // Bahdanau: scores on the VALU, the rest on the matrix pipe
  else if (d.attention == 1) launch_persist_wd<W, 1, 1>(d, Wcat, Wtok, bdec, Wcat1, bdec1, Nh, s);     // Bahdanau: one decoder cell, packed FMAs
  else if (d.mx_attention == 2) launch_persist_wd<W, 1, 3>(d, Wcat, Wtok, bdec, Wcat1, bdec1, Nh, s);  // Luong, attention and cell product on the matrix pipe
  else if (d.mx_attention) launch_persist_wd<W, 1, 2>(d, Wcat, Wtok, bdec, Wcat1, bdec1, Nh, s);  // Luong, scores and context on the matrix pipe
  else launch_persist_wd<W, 1, 0>(d, Wcat, Wtok, bdec, Wcat1, bdec1, Nh, s);
}
bool dec_persist_supported(const DecState& d) {
  const int att_form = d.attention == 1 ? (d.depth == 1 && d.mx_attention == 2 ? 4 : 1) : (d.depth <= 2 && d.mx_attention == 2 ? 3 : 0);
  if (sizeof(float) * PersistLds(d.W, d.depth > 1 ? 2 : 1, att_form).total + (att_form >= 3 ? MXC_STATIC_LDS : 10 * 1024) > 160 * 1024) return false;   // dynamic + static LDS
  return (d.attention == 0 || (d.attention == 1 && d.depth == 1)) && d.depth <= 2 && d.W <= (d.depth > 1 ? 5 : 8) && d.Tm <= 352 && !d.step_align && (!d.greedy || d.W == 1);
}
void launch_dec_persist(const DecState& d, const float* Wcat, const float* Wtok, const float* bdec,
                        const float* Wcat1, const float* bdec1, const float* Nh, hipStream_t s) {
  switch (d.W) {
    case 1: launch_persist_w<1>(d, Wcat, Wtok, bdec, Wcat1, bdec1, Nh, s); break;
    case 2: launch_persist_w<2>(d, Wcat, Wtok, bdec, Wcat1, bdec1, Nh, s); break;
    case 3: launch_persist_w<3>(d, Wcat, Wtok, bdec, Wcat1, bdec1, Nh, s); break;
    case 4: launch_persist_w<4>(d, Wcat, Wtok, bdec, Wcat1, bdec1, Nh, s); break;
    case 5: launch_persist_w<5>(d, Wcat, Wtok, bdec, Wcat1, bdec1, Nh, s); break;
    case 6: launch_persist_w<6>(d, Wcat, Wtok, bdec, Wcat1, bdec1, Nh, s); break;
    case 7: launch_persist_w<7>(d, Wcat, Wtok, bdec, Wcat1, bdec1, Nh, s); break;
    default: launch_persist_w<8>(d, Wcat, Wtok, bdec, Wcat1, bdec1, Nh, s); break;
  }
  // (S = max over chunk_steps is taken by k_dec_finalize)
}

void launch_dec_reduce_steps(const DecParts& p, hipStream_t s) {
  hipLaunchKernelGGL(k_dec_reduce_steps, dim3(1), dim3(64), 0, s, p);
}

void launch_input_mask(const float* raw, const float* ev, int B, int T_r, int T_e, float pad,
                       uint8_t* mask, hipStream_t s) {
  const int n = B * (T_r + T_e);
  hipLaunchKernelGGL(k_input_mask, dim3((n + 255) / 256), dim3(256), 0, s, raw, ev, B, T_r, T_e, pad, mask);
}
void launch_dec_init(const DecState& d, hipStream_t s) {
  const int n = max(d.B * d.W, d.L);
  hipLaunchKernelGGL(k_dec_init, dim3((n + 255) / 256), dim3(256), 0, s, d);
}
void launch_dec_cell(const DecState& d, int layer, const float* WcatT, const float* Wtok, const float* bias, int step,
                     hipStream_t s) {
  const int N = d.B * d.W;
  const size_t shm = sizeof(float) * CELL_LDS_FLOATS;
  hipLaunchKernelGGL(k_dec_cell, dim3(RV_U / 16, (N + CELL_ROWS - 1) / CELL_ROWS), dim3(512), shm, s, d, layer, WcatT, Wtok, bias, step);
}
template <int W, int TB, int TD>
static void launch_attend_wt(const DecState& d, int step, hipStream_t s) {
  const int TmP = (d.Tm + 3) & ~3;
  const size_t shm = sizeof(float) * AttLds(W, TmP, false).total;
  hipLaunchKernelGGL((k_dec_attend<W, TB, TD>), dim3(d.B), dim3(ATT_THREADS), shm, s, d, step);
}
template <int W>
static void launch_attend_w(const DecState& d, const float* WmemT, bool flash, int step, hipStream_t s) {
  if (flash) {
    if (d.attend_threads ? d.attend_threads == 256 : d.B > 320) {     // more chunks than CUs: two 256-thread workgroups per CU overlap each other's phases
      const size_t shm = sizeof(float) * AttLds(W, 0, true, 256).total;
      hipLaunchKernelGGL((k_dec_attend_flash<W, 256>), dim3(d.B), dim3(256), shm, s, d, WmemT, step);
    } else {
      const size_t shm = sizeof(float) * AttLds(W, 0, true, 512).total;
      hipLaunchKernelGGL((k_dec_attend_flash<W, 512>), dim3(d.B), dim3(512), shm, s, d, WmemT, step);
    }
    return;
  }
  if (d.Tm <= 64) launch_attend_wt<W, 2, 8>(d, step, s);
  else if (d.Tm <= 224) launch_attend_wt<W, 7, 28>(d, step, s);
  else launch_attend_wt<W, 11, 44>(d, step, s);
}
void launch_dec_attend(const DecState& d, const float* WmemT, bool flash, int step, hipStream_t s) {
  switch (d.W) {
    case 1: launch_attend_w<1>(d, WmemT, flash, step, s); break;
    case 2: launch_attend_w<2>(d, WmemT, flash, step, s); break;
    case 3: launch_attend_w<3>(d, WmemT, flash, step, s); break;
    case 4: launch_attend_w<4>(d, WmemT, flash, step, s); break;
    case 5: launch_attend_w<5>(d, WmemT, flash, step, s); break;
    case 6: launch_attend_w<6>(d, WmemT, flash, step, s); break;
    case 7: launch_attend_w<7>(d, WmemT, flash, step, s); break;
    default: launch_attend_w<8>(d, WmemT, flash, step, s); break;
  }
}
// Kernels that use more than the default dynamic-LDS limit opt in once per device (called by rv_create
// after hipSetDevice): sized for the largest shapes the library accepts (T_m <= 352).
template <int W>
static hipError_t configure_w() {
  constexpr int cap = 160 * 1024 - 10 * 1024;    // leave room for the kernels' static LDS
  hipError_t first = hipSuccess;
  auto opt = [&](const void* f, size_t bytes) {
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes < (size_t)cap ? bytes : (size_t)cap));
    if (e != hipSuccess && first == hipSuccess) first = e;
  };
  opt(reinterpret_cast<const void*>(&k_dec_attend<W, 2, 8>), sizeof(float) * AttLds(W, 64, false).total);
  opt(reinterpret_cast<const void*>(&k_dec_attend<W, 7, 28>), sizeof(float) * AttLds(W, 224, false).total);
  opt(reinterpret_cast<const void*>(&k_dec_attend<W, 11, 44>), sizeof(float) * AttLds(W, 352, false).total);
  opt(reinterpret_cast<const void*>(&k_dec_attend_flash<W, 256>), sizeof(float) * AttLds(W, 0, true, 256).total);
  opt(reinterpret_cast<const void*>(&k_dec_attend_flash<W, 512>), sizeof(float) * AttLds(W, 0, true, 512).total);
  opt(reinterpret_cast<const void*>(&k_dec_persist<W, 2, 1>), sizeof(float) * PersistLds(W, 1).total);
  opt(reinterpret_cast<const void*>(&k_dec_persist<W, 8, 1>), sizeof(float) * PersistLds(W, 1).total);
  opt(reinterpret_cast<const void*>(&k_dec_persist<W, 11, 1>), sizeof(float) * PersistLds(W, 1).total);
  opt(reinterpret_cast<const void*>(&k_dec_persist<W, 2, 1, 1>), sizeof(float) * PersistLds(W, 1, 1).total);
  opt(reinterpret_cast<const void*>(&k_dec_persist<W, 8, 1, 1>), sizeof(float) * PersistLds(W, 1, 1).total);
  opt(reinterpret_cast<const void*>(&k_dec_persist<W, 11, 1, 1>), sizeof(float) * PersistLds(W, 1, 1).total);
  opt(reinterpret_cast<const void*>(&k_dec_persist<W, 2, 1, 2>), sizeof(float) * PersistLds(W, 1).total);
  opt(reinterpret_cast<const void*>(&k_dec_persist<W, 8, 1, 2>), sizeof(float) * PersistLds(W, 1).total);
  opt(reinterpret_cast<const void*>(&k_dec_persist<W, 11, 1, 2>), sizeof(float) * PersistLds(W, 1).total);
  auto opt3 = [&](const void* f, size_t bytes) {       // (more dynamic LDS than `cap`: their static part is ~1 KB)
    const size_t cap3 = 160 * 1024 - MXC_STATIC_LDS;
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes < cap3 ? bytes : cap3));
    if (e != hipSuccess && first == hipSuccess) first = e;
  };
  opt3(reinterpret_cast<const void*>(&k_dec_persist<W, 2, 1, 3>), sizeof(float) * PersistLds(W, 1, 3).total);
  opt3(reinterpret_cast<const void*>(&k_dec_persist<W, 8, 1, 3>), sizeof(float) * PersistLds(W, 1, 3).total);
  opt3(reinterpret_cast<const void*>(&k_dec_persist<W, 11, 1, 3>), sizeof(float) * PersistLds(W, 1, 3).total);
  opt3(reinterpret_cast<const void*>(&k_dec_persist<W, 2, 1, 4>), sizeof(float) * PersistLds(W, 1, 4).total);
  opt3(reinterpret_cast<const void*>(&k_dec_persist<W, 8, 1, 4>), sizeof(float) * PersistLds(W, 1, 4).total);
  opt3(reinterpret_cast<const void*>(&k_dec_persist<W, 11, 1, 4>), sizeof(float) * PersistLds(W, 1, 4).total);
  if constexpr (W <= 5) {
    opt3(reinterpret_cast<const void*>(&k_dec_persist<W, 2, 2, 3>), sizeof(float) * PersistLds(W, 2, 3).total);
    opt3(reinterpret_cast<const void*>(&k_dec_persist<W, 8, 2, 3>), sizeof(float) * PersistLds(W, 2, 3).total);
    opt3(reinterpret_cast<const void*>(&k_dec_persist<W, 11, 2, 3>), sizeof(float) * PersistLds(W, 2, 3).total);
    opt(reinterpret_cast<const void*>(&k_dec_persist<W, 2, 2>), sizeof(float) * PersistLds(W, 2).total);
    opt(reinterpret_cast<const void*>(&k_dec_persist<W, 8, 2>), sizeof(float) * PersistLds(W, 2).total);
    opt(reinterpret_cast<const void*>(&k_dec_persist<W, 11, 2>), sizeof(float) * PersistLds(W, 2).total);
  }
  return first;
}
hipError_t configure_decode_kernels() {
  hipError_t first = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dec_cell), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(sizeof(float) * CELL_LDS_FLOATS));
  for (hipError_t e : {configure_w<1>(), configure_w<2>(), configure_w<3>(), configure_w<4>(),
                       configure_w<5>(), configure_w<6>(), configure_w<7>(), configure_w<8>()})
    if (e != hipSuccess && first == hipSuccess) first = e;
  return first;
}

void launch_dec_finalize(const DecState& d, int32_t* tokens, float* out2, hipStream_t s, const void* const* ptab) {
  hipLaunchKernelGGL(k_dec_finalize, dim3(d.B), dim3(64), 0, s, d, tokens, out2, ptab);
}
