/* ravvent_cpu.c -- fp32 C restatement of the Ravvent inference hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load the library built from this file.  PARITY UNPINNED (see the header
 * of oracle/ravvent_oracle.py: the reference's arithmetic lives in TensorFlow / TFA, which are
 * not installed here and hold no golden vectors for this path).  This file is validated against
 * the numpy fp64 oracle in tests/test_oracle.py.
 *
 * Structured like the reference's execution: per-timestep [rows,F]x[F,512] + [rows,128]x[128,512]
 * products inside a time loop (Keras RNN over LSTMCell, /root/reference/basecaller.py:19-32,
 * 48-59), per-step attention + log-softmax + top-k (TFA BeamSearchDecoder driven from
 * basecaller.py:306-313), OpenMP over the batch the way TF's intra-op pool splits it.
 * Units are fixed at 128 like every reference script (ravvent_performance_evaluator.py:92-93).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define U 128
#define G 512
#define E 256
#define RT 8 /* chunks per encoder tile */
#define MAXW 8
#define MAXV 8

typedef struct {
  int enc_depth, mode /*0 raw 1 event 2 joint*/, attention /*0 luong 1 bahdanau*/, vocab;
  int start_token, end_token, pad_token;
  float padding_value;
} RvoConfig;

typedef struct { const float *W, *Uk, *b; } LstmW;

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* z[r][n] += a[r][k] * Wm[k][n]   (multi-versioned: the .so is built in one container and runs on another host) */
__attribute__((target_clones("avx512f", "avx2", "default")))
static void gemm_acc(float* restrict z, const float* restrict a, const float* restrict Wm, int R, int K, int N, int lda) {
  for (int k = 0; k < K; ++k) {
    const float* restrict wr = Wm + (size_t)k * N;
    for (int r = 0; r < R; ++r) {
      const float av = a[(size_t)r * lda + k];
      float* restrict zr = z + (size_t)r * N;
      for (int n = 0; n < N; ++n) zr[n] += av * wr[n];
    }
  }
}

/* LSTMCell gate math: z holds x.W + h.U + b, order i f c~ o */
static void lstm_gates(const float* z, float* h, float* c, int R) {
  for (int r = 0; r < R; ++r) {
    const float* zr = z + (size_t)r * G;
    for (int j = 0; j < U; ++j) {
      const float ig = sigmoidf_(zr[j]), fg = sigmoidf_(zr[U + j]);
      const float gg = tanhf(zr[2 * U + j]), og = sigmoidf_(zr[3 * U + j]);
      const float c2 = fg * c[r * U + j] + ig * gg;
      c[r * U + j] = c2;
      h[r * U + j] = og * tanhf(c2);
    }
  }
}

/* One direction of one Bi-RNN layer over a tile of R chunks (Keras Bidirectional).
 * in: [B][T][F] rows b0..b0+R-1 ; out: [B][outT][256] at time offset t0, column offset dir*128 */
static void rnn_dir(const LstmW* w, int F, const float* in, int T, int b0, int R, int dir, float* h, float* c,
                    float* out, int outT, int t0, float* z) {
  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    for (int r = 0; r < R; ++r) memcpy(z + (size_t)r * G, w->b, sizeof(float) * G);
    gemm_acc(z, in + ((size_t)b0 * T + t) * F, w->W, R, F, G, T * F);
    gemm_acc(z, h, w->Uk, R, U, G, U);
    lstm_gates(z, h, c, R);
    for (int r = 0; r < R; ++r)
      memcpy(out + ((size_t)(b0 + r) * outT + t0 + t) * E + dir * U, h + r * U, sizeof(float) * U);
  }
}

static const float* bind_lstm(const float* p, LstmW* w, int F) {
  w->W = p; p += (size_t)F * G;
  w->Uk = p; p += (size_t)U * G;
  w->b = p; p += G;
  return p;
}

size_t rvo_weight_count(const RvoConfig* c) {
  size_t n = 0;
  for (int e = 0; e < 2; ++e)
    for (int l = 0; l < c->enc_depth; ++l) {
      const size_t F = l > 0 ? E : (e == 0 ? 1 : 5);
      n += 2 * (F * G + (size_t)U * G + G);
    }
  n += ((size_t)c->vocab + U) * G + (size_t)U * G + G;
  n += (size_t)E * U + (size_t)U * U + U + (size_t)(U + E) * U + (size_t)U * c->vocab + c->vocab;
  return n;
}

/* Encoder.call (basecaller.py:48-59) for all chunks: state chaining between layers. */
static void run_encoder(const LstmW (*lw)[2], int depth, int F0, const float* x, int B, int T, float* enc_out, int Tm,
                        int t_off) {
#pragma omp parallel
  {
    float* z = (float*)malloc(sizeof(float) * RT * G);
    float* st = (float*)malloc(sizeof(float) * 4 * RT * U);          /* h_f c_f h_b c_b */
    float* bufA = depth > 1 ? (float*)malloc(sizeof(float) * RT * T * E) : NULL;
    float* bufB = depth > 2 ? (float*)malloc(sizeof(float) * RT * T * E) : NULL;
#pragma omp for schedule(dynamic, 1)
    for (int b0 = 0; b0 < B; b0 += RT) {
      const int R = B - b0 < RT ? B - b0 : RT;
      memset(st, 0, sizeof(float) * 4 * RT * U);                      /* layer 0 starts at zeros */
      const float* in = x + (size_t)b0 * T * F0;
      int F = F0;
      for (int l = 0; l < depth; ++l) {
        const int last = l == depth - 1;
        float* out = last ? enc_out + (size_t)b0 * Tm * E : ((l & 1) ? bufB : bufA);
        /* tile-local views: rows are 0..R-1 of `in` / `out` */
        rnn_dir(&lw[l][0], F, in, T, 0, R, 0, st, st + RT * U, out, last ? Tm : T, last ? t_off : 0, z);
        rnn_dir(&lw[l][1], F, in, T, 0, R, 1, st + 2 * RT * U, st + 3 * RT * U, out, last ? Tm : T, last ? t_off : 0, z);
        in = out; F = E;
      }
    }
    free(z); free(st); free(bufA); free(bufB);
  }
}

/* Full path.  greedy=0: Basecaller.beam_search_prediction (basecaller.py:296-315) -> tokens
 * [B][L-1], out2 = scores [B][L-1].  greedy=1: greedy_search_prediction (:317-330) -> tokens,
 * out2 = logits [B][L-1][V].  Returns S (decode steps executed) or <0. */
int rvo_run(const RvoConfig* cfg, const float* blob, const float* raw, const float* ev, int B, int T_r, int T_e, int W,
            int L, int greedy, int32_t* tokens, float* out2, int nthreads) {
  const int V = cfg->vocab;
  if (V > MAXV || W > MAXW || W < 1 || cfg->enc_depth < 1 || cfg->enc_depth > 8) return -1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  if (cfg->mode == 1) T_r = 0;
  if (cfg->mode == 0) T_e = 0;
  if (greedy) W = 1;
  const int Tm = T_r + T_e, steps = L - 1;
  if (B <= 0 || steps <= 0) return 0;

  LstmW enc[2][8][2], dec;
  const float* p = blob;
  for (int e = 0; e < 2; ++e)
    for (int l = 0; l < cfg->enc_depth; ++l)
      for (int d = 0; d < 2; ++d) p = bind_lstm(p, &enc[e][l][d], l > 0 ? E : (e == 0 ? 1 : 5));
  p = bind_lstm(p, &dec, V + U);
  const float* W_mem = p; p += (size_t)E * U;
  const float* W_q = p; p += (size_t)U * U;
  const float* v_att = p; p += U;
  const float* W_att = p; p += (size_t)(U + E) * U;
  const float* W_fc = p; p += (size_t)U * V;
  const float* b_fc = p;

  float* enc_out = (float*)malloc(sizeof(float) * (size_t)B * Tm * E);
  float* keys = (float*)malloc(sizeof(float) * (size_t)B * Tm * U);
  uint8_t* mask = (uint8_t*)malloc((size_t)B * Tm);
  const int N = B * W;
  float* hS = (float*)calloc((size_t)N * U, sizeof(float));
  float* cS = (float*)calloc((size_t)N * U, sizeof(float));
  float* aS = (float*)calloc((size_t)N * U, sizeof(float));
  int* tok = (int*)malloc(sizeof(int) * N);
  float* lprob = (float*)malloc(sizeof(float) * N);
  uint8_t* fin = (uint8_t*)calloc(N, 1);
  int* len = (int*)calloc(N, sizeof(int));
  int* ids = (int*)malloc(sizeof(int) * (size_t)steps * N);
  int* par = (int*)malloc(sizeof(int) * (size_t)steps * N);
  float* ssc = (float*)malloc(sizeof(float) * (size_t)steps * N);
  float* slg = greedy ? (float*)malloc(sizeof(float) * (size_t)steps * N * V) : NULL;

  /* _encode_input (basecaller.py:395-416): masks from the raw inputs, raw part then event part */
  for (int b = 0; b < B; ++b) {
    for (int t = 0; t < T_r; ++t) mask[(size_t)b * Tm + t] = raw[(size_t)b * T_r + t] != cfg->padding_value;
    for (int t = 0; t < T_e; ++t) {
      const float* e5 = ev + ((size_t)b * T_e + t) * 5;
      int ok = 1;
      for (int f = 0; f < 5; ++f) ok &= e5[f] != cfg->padding_value;
      mask[(size_t)b * Tm + T_r + t] = (uint8_t)ok;
    }
  }
  if (T_r > 0) run_encoder(enc[0], cfg->enc_depth, 1, raw, B, T_r, enc_out, Tm, 0);
  if (T_e > 0) run_encoder(enc[1], cfg->enc_depth, 5, ev, B, T_e, enc_out, Tm, T_r);

  /* setup_memory (basecaller.py:303): keys = (memory*mask).W_mem */
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < Tm; ++t) {
      float* k = keys + ((size_t)b * Tm + t) * U;
      memset(k, 0, sizeof(float) * U);
      if (mask[(size_t)b * Tm + t]) gemm_acc(k, enc_out + ((size_t)b * Tm + t) * E, W_mem, 1, E, U, E);
    }

  for (int n = 0; n < N; ++n) { tok[n] = cfg->start_token; lprob[n] = (n % W) == 0 ? 0.f : -INFINITY; }

  int S = 0;
  for (int step = 0; step < steps; ++step) {
    int all = 1;
    for (int n = 0; n < N; ++n) all &= fin[n];
    if (all) break;
#pragma omp parallel
    {
      float z[MAXW * G], hn[MAXW * U], cn[MAXW * U], an[MAXW * U], hc[MAXW * (U + E)], lg[MAXW * MAXV];
      float tot[MAXW * MAXV], pq[U];
      float* sc = (float*)malloc(sizeof(float) * Tm);
#pragma omp for schedule(static)
      for (int b = 0; b < B; ++b) {
        const size_t r0 = (size_t)b * W;
        /* AttentionWrapper step: cell input = [one_hot(tok) ; attention] */
        for (int w = 0; w < W; ++w) {
          float* zr = z + w * G;
          const float* wt = dec.W + (size_t)tok[r0 + w] * G;
          for (int n = 0; n < G; ++n) zr[n] = dec.b[n] + wt[n];
        }
        gemm_acc(z, aS + r0 * U, dec.W + (size_t)V * G, W, U, G, U);
        gemm_acc(z, hS + r0 * U, dec.Uk, W, U, G, U);
        memcpy(hn, hS + r0 * U, sizeof(float) * W * U);
        memcpy(cn, cS + r0 * U, sizeof(float) * W * U);
        lstm_gates(z, hn, cn, W);
        for (int w = 0; w < W; ++w) {
          const float* q = hn + w * U;
          if (cfg->attention == 1) { memset(pq, 0, sizeof pq); gemm_acc(pq, q, W_q, 1, U, U, U); }
          float m = -INFINITY;
          for (int t = 0; t < Tm; ++t) {
            const float* k = keys + ((size_t)b * Tm + t) * U;
            float s = 0.f;
            if (cfg->attention == 1) for (int j = 0; j < U; ++j) s += v_att[j] * tanhf(k[j] + pq[j]);
            else for (int j = 0; j < U; ++j) s += q[j] * k[j];
            s = mask[(size_t)b * Tm + t] ? s : -INFINITY;
            sc[t] = s;
            m = fmaxf(m, s);
          }
          float sum = 0.f;
          for (int t = 0; t < Tm; ++t) { sc[t] = expf(sc[t] - m); sum += sc[t]; }
          float* hcw = hc + w * (U + E);
          memcpy(hcw, q, sizeof(float) * U);
          memset(hcw + U, 0, sizeof(float) * E);
          for (int t = 0; t < Tm; ++t) {
            const float a = sc[t] / sum;
            if (a == 0.f) continue;
            const float* v = enc_out + ((size_t)b * Tm + t) * E;
            for (int i = 0; i < E; ++i) hcw[U + i] += a * v[i];
          }
        }
        memset(an, 0, sizeof(float) * W * U);
        gemm_acc(an, hc, W_att, W, U + E, U, U + E);
        for (int w = 0; w < W; ++w)
          for (int v = 0; v < V; ++v) {
            float s = 0.f;
            for (int k = 0; k < U; ++k) s += an[w * U + k] * W_fc[k * V + v];
            lg[w * V + v] = s + b_fc[v];
          }
        const size_t o = (size_t)step * N + r0;
        if (greedy) {
          int best = 0;
          for (int v = 1; v < V; ++v) if (lg[v] > lg[best]) best = v;
          memcpy(slg + o * V, lg, sizeof(float) * V);
          ids[o] = best; par[o] = 0; ssc[o] = lg[best];
          fin[r0] = fin[r0] || best == cfg->end_token;
          tok[r0] = best;
          memcpy(hS + r0 * U, hn, sizeof(float) * U);
          memcpy(cS + r0 * U, cn, sizeof(float) * U);
          memcpy(aS + r0 * U, an, sizeof(float) * U);
          continue;
        }
        /* _beam_search_step */
        for (int w = 0; w < W; ++w) {
          float m = lg[w * V];
          for (int v = 1; v < V; ++v) m = fmaxf(m, lg[w * V + v]);
          float s = 0.f;
          for (int v = 0; v < V; ++v) s += expf(lg[w * V + v] - m);
          const float lse = logf(s);
          for (int v = 0; v < V; ++v) {
            const float lp = fin[r0 + w] ? (v == cfg->end_token ? 0.f : -FLT_MAX) : (lg[w * V + v] - m) - lse;
            tot[w * V + v] = lprob[r0 + w] + lp;
          }
        }
        uint64_t taken = 0;
        int nw[MAXW], np_[MAXW], nl[MAXW]; float nv[MAXW]; uint8_t nf[MAXW];
        for (int k = 0; k < W; ++k) {
          int best = -1;
          for (int cnd = 0; cnd < W * V; ++cnd) {
            if (taken >> cnd & 1) continue;
            if (best < 0 || tot[cnd] > tot[best]) best = cnd;
          }
          taken |= (uint64_t)1 << best;
          nw[k] = best % V; np_[k] = best / V; nv[k] = tot[best];
          const int pf = fin[r0 + np_[k]];
          nf[k] = (uint8_t)(pf || nw[k] == cfg->end_token);
          nl[k] = len[r0 + np_[k]] + (pf ? 0 : 1);
        }
        for (int k = 0; k < W; ++k) {
          ids[o + k] = nw[k]; par[o + k] = np_[k]; ssc[o + k] = nv[k];
          tok[r0 + k] = nw[k]; lprob[r0 + k] = nv[k]; fin[r0 + k] = nf[k]; len[r0 + k] = nl[k];
          memcpy(hS + (r0 + k) * U, hn + np_[k] * U, sizeof(float) * U);
          memcpy(cS + (r0 + k) * U, cn + np_[k] * U, sizeof(float) * U);
          memcpy(aS + (r0 + k) * U, an + np_[k] * U, sizeof(float) * U);
        }
      }
      free(sc);
    }
    S = step + 1;
  }

  /* finalize: gather_tree beam 0 + predicted_ids[:,:,0] / scores[:,:,0] (basecaller.py:315) */
  for (int b = 0; b < B; ++b) {
    int32_t* tk = tokens + (size_t)b * steps;
    if (greedy) {
      for (int s = 0; s < steps; ++s) {
        tk[s] = s < S ? ids[(size_t)s * N + b] : cfg->pad_token;
        for (int v = 0; v < V; ++v) out2[((size_t)b * steps + s) * V + v] = s < S ? slg[((size_t)s * N + b) * V + v] : 0.f;
      }
      continue;
    }
    int maxlen = 0;
    for (int w = 0; w < W; ++w) if (len[(size_t)b * W + w] > maxlen) maxlen = len[(size_t)b * W + w];
    const int Lb = S < maxlen ? S : maxlen;
    for (int s = 0; s < steps; ++s) tk[s] = s < S ? cfg->end_token : cfg->pad_token;
    if (Lb > 0) {
      tk[Lb - 1] = ids[(size_t)(Lb - 1) * N + (size_t)b * W];
      int pp = par[(size_t)(Lb - 1) * N + (size_t)b * W];
      for (int t = Lb - 2; t >= 0; --t) { tk[t] = ids[(size_t)t * N + (size_t)b * W + pp]; pp = par[(size_t)t * N + (size_t)b * W + pp]; }
      int done = 0;
      for (int t = 0; t < Lb; ++t) { if (done) tk[t] = cfg->end_token; else if (tk[t] == cfg->end_token) done = 1; }
    }
    for (int s = 0; s < steps; ++s) out2[(size_t)b * steps + s] = s < S ? ssc[(size_t)s * N + (size_t)b * W] : 0.f;
  }
  free(enc_out); free(keys); free(mask); free(hS); free(cS); free(aS); free(tok); free(lprob); free(fin); free(len);
  free(ids); free(par); free(ssc); free(slg);
  return S;
}

int rvo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
