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

__global__ void k_dec_gates(DecState d, int step) {
  if (step > 0 && d.nfin[step - 1] >= d.B) return;
  const int N = d.B * d.W;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * RV_U) return;
  const int n = idx / RV_U, j = idx % RV_U;
  const float* z = d.z + (size_t)n * RV_G + j;
  const float ig = rv_sigmoid(z[0]), fg = rv_sigmoid(z[RV_U]);
  const float gg = rv_tanh(z[2 * RV_U]), og = rv_sigmoid(z[3 * RV_U]);
  const float c2 = fmaf(fg, d.c[idx], ig * gg);
  d.c_new[idx] = c2;
  d.h_new[idx] = og * rv_tanh(c2);
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

constexpr int WB = RV_MAX_BEAM;

__global__ __launch_bounds__(256) void k_dec_attend(DecState d, int step) {
  extern __shared__ __align__(16) float dsm[];
  const int Tm = d.Tm, W = d.W, V = d.V;
  const int TmP = (Tm + 3) & ~3;
  float* sc = dsm;                 // [WB][TmP]   scores, then exp()
  float* al = dsm + WB * TmP;      // [TmP][WB]   alignments, beam-minor for the context sweep
  __shared__ __align__(16) float q[WB][RV_U];
  __shared__ __align__(16) float pq[WB][RV_U];
  __shared__ __align__(16) float ctx[WB][RV_E];
  __shared__ __align__(16) float part[WB][2][RV_U];
  __shared__ __align__(16) float att[WB][RV_U];
  __shared__ float lg[WB][RV_MAX_VOCAB];
  __shared__ float tot[WB * RV_MAX_VOCAB];
  __shared__ int s_parent[WB];

  const int b = blockIdx.x, tid = threadIdx.x;
  if (step > 0 && d.nfin[step - 1] >= d.B) {      // whole batch finished earlier: propagate
    if (tid == 0) atomicAdd(&d.nfin[step], 1);
    return;
  }
  const size_t row0 = (size_t)b * W;

  // ---- A: query = cell output h of every beam
  for (int i = tid; i < W * RV_U; i += 256) q[i >> 7][i & 127] = d.h_new[row0 * RV_U + i];
  __syncthreads();
  if (d.attention == 1) {   // Bahdanau: processed query = q . W_q
    const int jj = tid & 127, half = tid >> 7;
    float acc[WB];
#pragma unroll
    for (int w = 0; w < WB; ++w) acc[w] = 0.f;
    for (int k = half * 64; k < half * 64 + 64; ++k) {
      const float wq = d.W_q[k * RV_U + jj];
#pragma unroll
      for (int w = 0; w < WB; ++w) acc[w] = fmaf(q[w][k], wq, acc[w]);
    }
#pragma unroll
    for (int w = 0; w < WB; ++w) part[w][half][jj] = acc[w];
    __syncthreads();
    if (tid < RV_U)
#pragma unroll
      for (int w = 0; w < WB; ++w) pq[w][tid] = part[w][0][tid] + part[w][1][tid];
    __syncthreads();
  }

  // ---- B: scores.  16 lanes share one memory step t (8 key columns each), 16 steps in flight.
  {
    const int sub = tid & 15, grp = tid >> 4;
    float qr[WB][8], vr[8];
#pragma unroll
    for (int w = 0; w < WB; ++w)
#pragma unroll
      for (int i = 0; i < 8; ++i) qr[w][i] = d.attention == 1 ? pq[w][8 * sub + i] : q[w][8 * sub + i];
#pragma unroll
    for (int i = 0; i < 8; ++i) vr[i] = d.attention == 1 ? d.v_att[8 * sub + i] : 0.f;
    for (int t = grp; t < Tm; t += 16) {
      const float4* kp = reinterpret_cast<const float4*>(d.keys + ((size_t)b * Tm + t) * RV_U + 8 * sub);
      const float4 k0 = kp[0], k1 = kp[1];
      const float kk[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
      const bool live = d.mask[(size_t)b * Tm + t] != 0;
#pragma unroll
      for (int w = 0; w < WB; ++w) {
        if (w >= W) break;
        float p = 0.f;
        if (d.attention == 1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) p = fmaf(vr[i], tanhf(kk[i] + qr[w][i]), p);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) p = fmaf(kk[i], qr[w][i], p);
        }
        p += __shfl_xor(p, 8, 16); p += __shfl_xor(p, 4, 16);
        p += __shfl_xor(p, 2, 16); p += __shfl_xor(p, 1, 16);
        if (sub == 0) sc[w * TmP + t] = live ? p : -INFINITY;    // _maybe_mask_score
      }
    }
  }
  __syncthreads();

  // ---- C: softmax over T_m, one wave per beam
  {
    const int lane = tid & 63, wv = tid >> 6;
    for (int w = wv; w < W; w += 4) {
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
    for (int i = tid; i < Tm * WB; i += 256)
      if ((i & (WB - 1)) >= W) al[i] = 0.f;
  }
  __syncthreads();

  // ---- D: context = sum_t alpha_t * values_t ; thread = output column
  {
    float acc[WB];
#pragma unroll
    for (int w = 0; w < WB; ++w) acc[w] = 0.f;
    const float* vp = d.values + (size_t)b * Tm * RV_E + tid;
#pragma unroll 4
    for (int t = 0; t < Tm; ++t) {
      const float v = vp[(size_t)t * RV_E];
      const float4 a0 = *reinterpret_cast<const float4*>(&al[t * WB]);
      const float4 a1 = *reinterpret_cast<const float4*>(&al[t * WB + 4]);
      acc[0] = fmaf(a0.x, v, acc[0]); acc[1] = fmaf(a0.y, v, acc[1]);
      acc[2] = fmaf(a0.z, v, acc[2]); acc[3] = fmaf(a0.w, v, acc[3]);
      acc[4] = fmaf(a1.x, v, acc[4]); acc[5] = fmaf(a1.y, v, acc[5]);
      acc[6] = fmaf(a1.z, v, acc[6]); acc[7] = fmaf(a1.w, v, acc[7]);
    }
#pragma unroll
    for (int w = 0; w < WB; ++w) ctx[w][tid] = acc[w];
  }
  __syncthreads();

  // ---- E: attention = [h ; context] . W_att   (Dense, no bias, no activation)
  {
    const int dd = tid & 127, half = tid >> 7;
    float acc[WB];
#pragma unroll
    for (int w = 0; w < WB; ++w) acc[w] = 0.f;
    for (int k = half * 192; k < half * 192 + 192; ++k) {
      const float wa = d.W_att[k * RV_U + dd];
#pragma unroll
      for (int w = 0; w < WB; ++w) {
        const float hv = k < RV_U ? q[w][k] : ctx[w][k - RV_U];
        acc[w] = fmaf(hv, wa, acc[w]);
      }
    }
#pragma unroll
    for (int w = 0; w < WB; ++w) part[w][half][dd] = acc[w];
  }
  __syncthreads();
  if (tid < RV_U)
#pragma unroll
    for (int w = 0; w < WB; ++w) att[w][tid] = part[w][0][tid] + part[w][1][tid];
  __syncthreads();

  // ---- F: logits = attention . W_fc + b_fc
  if (tid < W * V) {
    const int w = tid / V, v = tid % V;
    float acc = 0.f;
    for (int k = 0; k < RV_U; ++k) acc = fmaf(att[w][k], d.W_fc[k * V + v], acc);
    acc += d.b_fc[v];
    lg[w][v] = acc;
    if (d.step_logits) d.step_logits[(((size_t)step * d.B + b) * W + w) * V + v] = acc;
  }
  __syncthreads();

  // ---- G: sampler / beam step (tiny: W*V <= 64 candidates)
  if (tid == 0) {
    const size_t o = ((size_t)step * d.B + b) * W;
    if (d.greedy) {
      int best = 0;
      for (int v = 1; v < V; ++v) if (lg[0][v] > lg[0][best]) best = v;   // first max on ties
      const bool fin = d.finished[row0] || best == d.end_token;
      d.step_ids[o] = best; d.parent_ids[o] = 0; d.step_scores[o] = lg[0][best];
      d.tok[row0] = best;
      d.finished[row0] = fin;
      s_parent[0] = 0;
      if (fin) atomicAdd(&d.nfin[step], 1);
    } else {
      for (int w = 0; w < W; ++w) {
        float m = lg[w][0];
        for (int v = 1; v < V; ++v) m = fmaxf(m, lg[w][v]);
        float s = 0.f;
        for (int v = 0; v < V; ++v) s += expf(lg[w][v] - m);
        const float lse = logf(s);
        const bool fin = d.finished[row0 + w] != 0;
        const float lpw = d.log_probs[row0 + w];
        for (int v = 0; v < V; ++v) {
          const float lp = fin ? (v == d.end_token ? 0.f : -FLT_MAX) : (lg[w][v] - m) - lse;
          tot[w * V + v] = lpw + lp;
        }
      }
      unsigned long long taken = 0ull;
      int word[WB], par[WB], nlen[WB]; float val[WB]; bool nfin[WB];
      bool all = true;
      for (int k = 0; k < W; ++k) {
        int best = -1;
        for (int cnd = 0; cnd < W * V; ++cnd) {
          if (taken >> cnd & 1ull) continue;
          if (best < 0 || tot[cnd] > tot[best]) best = cnd;    // ties -> lower index
        }
        taken |= 1ull << best;
        word[k] = best % V; par[k] = best / V; val[k] = tot[best];
        const bool pf = d.finished[row0 + par[k]] != 0;
        nfin[k] = pf || word[k] == d.end_token;
        nlen[k] = d.lengths[row0 + par[k]] + (pf ? 0 : 1);
        all = all && nfin[k];
      }
      for (int k = 0; k < W; ++k) {
        d.step_ids[o + k] = word[k]; d.parent_ids[o + k] = par[k]; d.step_scores[o + k] = val[k];
        d.tok[row0 + k] = word[k];
        d.log_probs[row0 + k] = val[k];
        d.finished[row0 + k] = nfin[k];
        d.lengths[row0 + k] = nlen[k];
        s_parent[k] = par[k];
      }
      if (all) atomicAdd(&d.nfin[step], 1);
    }
  }
  __syncthreads();

  // ---- H: next-step state, gathered by parent beam: xh = [attention | h], c
  for (int i = tid; i < W * RV_U; i += 256) {
    const int w = i >> 7, e = i & 127, p = s_parent[w];
    d.xh[(row0 + w) * RV_E + e] = att[p][e];
    d.xh[(row0 + w) * RV_E + RV_U + e] = q[p][e];
    d.c[(row0 + w) * RV_U + e] = d.c_new[(row0 + p) * RV_U + e];
  }
}

__global__ void k_dec_finalize(DecState d, int32_t* tokens, float* out2) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.B) return;
  const int steps = d.L - 1;
  int S = steps;
  for (int s = 0; s < steps; ++s)
    if (d.nfin[s] >= d.B) { S = s + 1; break; }
  if (b == 0) *d.S_dev = S;
  int32_t* tk = tokens + (size_t)b * steps;
  const int W = d.W, V = d.V;
  if (d.greedy) {
    float* lo = out2 + (size_t)b * steps * V;
    for (int s = 0; s < steps; ++s) {
      tk[s] = s < S ? d.step_ids[(size_t)s * d.B + b] : d.pad_token;
      for (int v = 0; v < V; ++v)
        lo[s * V + v] = s < S ? d.step_logits[((size_t)s * d.B + b) * V + v] : 0.f;
    }
    return;
  }
  // gather_tree for beam 0
  int maxlen = 0;
  for (int w = 0; w < W; ++w) maxlen = max(maxlen, d.lengths[(size_t)b * W + w]);
  const int Lb = min(S, maxlen);
  for (int s = 0; s < steps; ++s) tk[s] = s < S ? d.end_token : d.pad_token;
  if (Lb > 0) {
    tk[Lb - 1] = d.step_ids[((size_t)(Lb - 1) * d.B + b) * W];
    int p = d.parent_ids[((size_t)(Lb - 1) * d.B + b) * W];
    for (int t = Lb - 2; t >= 0; --t) {
      tk[t] = d.step_ids[((size_t)t * d.B + b) * W + p];
      p = d.parent_ids[((size_t)t * d.B + b) * W + p];
    }
    bool done = false;
    for (int t = 0; t < Lb; ++t) {
      if (done) tk[t] = d.end_token;
      else if (tk[t] == d.end_token) done = true;
    }
  }
  float* sco = out2 + (size_t)b * steps;
  for (int s = 0; s < steps; ++s) sco[s] = s < S ? d.step_scores[((size_t)s * d.B + b) * W] : 0.f;
}

}  // namespace

void launch_input_mask(const float* raw, const float* ev, int B, int T_r, int T_e, float pad,
                       uint8_t* mask, hipStream_t s) {
  const int n = B * (T_r + T_e);
  hipLaunchKernelGGL(k_input_mask, dim3((n + 255) / 256), dim3(256), 0, s, raw, ev, B, T_r, T_e, pad, mask);
}
void launch_dec_init(const DecState& d, hipStream_t s) {
  const int n = max(d.B * d.W, d.L);
  hipLaunchKernelGGL(k_dec_init, dim3((n + 255) / 256), dim3(256), 0, s, d);
}
void launch_dec_gates(const DecState& d, int step, hipStream_t s) {
  const int n = d.B * d.W * RV_U;
  hipLaunchKernelGGL(k_dec_gates, dim3((n + 255) / 256), dim3(256), 0, s, d, step);
}
void launch_dec_attend(const DecState& d, int step, hipStream_t s) {
  const int TmP = (d.Tm + 3) & ~3;
  const size_t shm = sizeof(float) * 2 * WB * TmP;
  hipLaunchKernelGGL(k_dec_attend, dim3(d.B), dim3(256), shm, s, d, step);
}
void launch_dec_finalize(const DecState& d, int32_t* tokens, float* out2, hipStream_t s) {
  hipLaunchKernelGGL(k_dec_finalize, dim3((d.B + 63) / 64), dim3(64), 0, s, d, tokens, out2);
}
