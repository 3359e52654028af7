// Shared declarations of the gfx950 kernels behind libravvent_hip.so.
// Units are fixed at u = d = 128 (every reference script uses 128:
// /root/reference/ravvent_performance_evaluator.py:92-93, ravvent.py:14-15); the recurrence
// kernel's register tiling is built around that number.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RV_U 128          // LSTM units per direction / decoder units
#define RV_G 512          // 4 gates x RV_U
#define RV_E 256          // encoder output width 2u
#define RV_MAX_BEAM 8
#define RV_MAX_VOCAB 8

// ---------------------------------------------------------------- K1: BiLSTM recurrence
struct RecArgs {
  const float* x;        // F>0: chunk input [B,T,F];  F==0: pre-projected xw [B,T,2,512] (bias folded)
  const float* W[2];     // F>0: input kernel [F,512] per direction
  const float* bias[2];  // F>0: [512] per direction
  const float* U[2];     // recurrent kernel [128,512] per direction (blob order)
  const float* Up[2];    // the same kernel in the recurrence's register order: [32 i][512 threads] float4 = (slot 0..3 of k = 32 kq + i), slot r = gate (kq + r) & 3 of unit j; thread = 4 j + kq
  const float* Wp[2];    // fused-projection kernel only: input kernel [256,512] as MFMA B fragments [32 tiles][16 k-groups][64 lanes][4]
  const uint16_t* Wsb[2];  // the same kernel as three bf16 parts for the split-bf16 projection, [32 tiles][3 parts][8 k-steps][64 lanes][8]; null = f32 MFMA
  const uint16_t* Wh[2];   // ... as two f16 parts of the column-scaled kernel, [32 tiles][2 parts][8 k-steps][64 lanes][8], then 512 floats 2^-14 / s_n; takes precedence over Wsb
  const uint16_t* Ua[2];   // matrix-pipe recurrence (lstm_mx.hip): U^T as MFMA A fragments, [8 waves][4 gates][4 k-steps][2 parts][64 lanes][8 f16]
                           // of the row-scaled kernel, then 512 floats 2^-14 / s_r (RV_UA_SLOT uint16 per direction)
  const float* h0[2];    // initial states [B,128] or nullptr (zeros)
  const float* c0[2];
  float* hT[2];          // final states [B,128]
  float* cT[2];
  float* out;            // [B,out_T,256]; direction d writes columns [128d,128d+128) at time out_t0+t
  int out_T, out_t0;
  int B, T;
  long long* dbg_ts;     // diagnostic builds (-DRV_REC_STAMPS, env RV_REC_STAMPS): [24] per-wave {busy, barrier-wait} cycle sums of workgroup (0,0)
  int tail_wave;         // layer 0 (F > 0), rows_per_block >= 2: the cell update runs on a ninth wave (k_lstm_rec_tw)
  int dbg_role;          // timing probe only (RV_DBG_ROLE): 1 = projection waves skip their math, 2 = recurrence waves skip theirs
  uint8_t* mask;         // matrix-pipe raw layer 0 only: also leaves utils.input_mask of its chunks at mask[b * mask_T + mask_t0 + t]
  int mask_T, mask_t0;   //   (the window is in LDS anyway; saves the slab a launch).  null = not asked for
  float pad;
  const void* const* ptab;   // matrix-pipe raw layer 0 only: non-null = read the chunk inputs' address from ptab[RV_PTAB_RAW] instead of `x`
};
// Caller pointers of a slab call, read by the kernels through a small table in mapped pinned memory when the slab replays from a
// hipGraph whose kernel arguments are frozen (ravvent_hip.cpp, option "slab_graph": one graph per slab context and call shape): the
// only addresses that change from call to call.
enum { RV_PTAB_RAW = 0, RV_PTAB_EVENT = 1, RV_PTAB_TOKENS = 2, RV_PTAB_OUT2 = 3, RV_PTAB_N = 4 };
// F in {0,1,5}; rows_per_block in {1,2,4,8}
void launch_lstm_rec(const RecArgs& a, int F, int rows_per_block, hipStream_t s);
// layers >= 1 with x . W + b computed inside the kernel (MFMA waves beside the recurrence waves): a.x = [B,T,256] activations,
// a.Wp / a.bias per direction
void launch_lstm_rec_proj(const RecArgs& a, int rows_per_block, hipStream_t s);
// Matrix-pipe recurrence, RV_MX_ROWS chunks of one direction per workgroup (lstm_mx.hip).  F == 1: raw layer 0 (a.x = chunk inputs
// [B,T,1], a.W / a.bias per direction); F == 0: a.x = pre-projected inputs xw [B,T,2,512] (bias folded).  Needs a.Ua.
#define RV_MX_ROWS 16
#define RV_UA_SLOT ((size_t)2 * RV_U * RV_G + 2 * RV_G)
void launch_lstm_rec_mx(const RecArgs& a, int F, hipStream_t s, bool rows8 = false);   // rows8: eight chunks per workgroup (latency form)
bool lstm_rec_mx_window_fits(int T, int F = 1);   // layer 0 with its input projection in the lane: do the input windows of a workgroup's chunks fit in LDS?
hipError_t configure_mx_kernels();
// xw [rows,2,512] = x [rows,F] . W_dir [F,512] + b_dir for a layer-0 encoder with F = 5 (or 1) input features, both directions
// mask != null: also writes utils.input_mask of the rows, mask[(r / T) * mask_T + mask_t0 + r % T] = all(x[r, :] != pad)
// xtab != null: x is read from xtab[RV_PTAB_EVENT] (F == 5) / xtab[RV_PTAB_RAW] (F == 1) on the device instead
void launch_inproj_small(const float* x, int rows, int F, const float* W0, const float* b0, const float* W1, const float* b1, float* xw,
                         uint8_t* mask, int T, int mask_T, int mask_t0, float pad, hipStream_t s, const void* const* xtab = nullptr);
hipError_t configure_rec_kernels();   // dynamic-LDS opt-in of the recurrence kernels; first error or hipSuccess
// layer 0 stages its chunks' whole input windows in LDS: does a window of T steps x F features fit with that many rows per workgroup?
bool lstm_rec_window_fits(int F, int rows_per_block, int T);

// ---------------------------------------------------------------- K0/K2: fp32 MFMA GEMM
struct GemmArgs {
  const float* A; int lda;     // [M,K]
  const float* Bm; int ldb;    // [K,N]
  float* C; int ldc;           // [M,N]
  int M, N, K;                 // K % 16 == 0, N % 64 == 0
  const float* bias;           // [N] or nullptr
  const uint8_t* row_mask;     // [M] or nullptr: masked-off rows are written as 0
  const int* gather_idx;       // [M] or nullptr: adds gather_tab[gather_idx[m]*ld_tab + n]
  const float* gather_tab; int ld_tab;
  // optional second problem sharing A (blockIdx.z == 1): the other LSTM direction of the same layer
  const float* Bm1; const float* bias1; float* C1;
  int xcd_remap;               // 1: 1-D grid, the workgroups sharing an A tile are dealt to ONE XCD (L2 reuse)
  const int* skip_flag; int skip_when;   // if skip_flag && *skip_flag >= skip_when: kernel exits
};
void launch_gemm_f32(const GemmArgs& a, bool small_tile, hipStream_t s);
// C[M,256] = A[M,256] . Wmp[256,256] on split-f16 MFMAs; img = [8 k-steps][16 tiles][2 parts][64 lanes][8 f16] of the column-scaled
// kernel, then 256 floats 2^-14 / s_n (RV_WMP16_SLOT uint16).  Precondition: |A| <= 1 (rows of an LSTM layer's output).
#define RV_WMP16_SLOT ((size_t)2 * RV_E * RV_E + 2 * RV_E)
void launch_gemm_mem_split(const float* A, int M, const uint16_t* img, float* C, hipStream_t s);
// The same kernel over `ncb` column blocks of 256: C[M, ldc] (columns 256 cb ..) = A[M,256] . W_cb[256,256] (+ bias[256 cb ..]);
// img = [ncb][8 k-steps][16 tiles][2 parts][64 lanes][8 f16], then 256 ncb floats 2^-14 / s_n.  A workgroup takes the ncb blocks of a
// row tile one after the other, so A comes from HBM once.  The input projection of encoder layers >= 1 for the matrix-pipe
// recurrence: ncb = 4 (two directions x 512 gate columns), ldc = 1024, bias = [b_fwd | b_bwd].
#define RV_WX16_SLOT ((size_t)2 * RV_E * 2 * RV_G + 2 * 2 * RV_G)
hipError_t configure_gemm_kernels();   // dynamic-LDS opt-in of the weight-stationary split GEMM
void launch_gemm_split_blocks(const float* A, int M, const uint16_t* img, int ncb, const float* bias, float* C, int ldc, hipStream_t s);

// ---------------------------------------------------------------- small encoder-side kernels
void launch_input_mask(const float* raw, const float* ev, int B, int T_r, int T_e, float pad,
                       uint8_t* mask /*[B,T_r+T_e]*/, hipStream_t s);

// ---------------------------------------------------------------- decoder
struct DecState {
  // per-call shapes
  int B, W, Tm, V, L;
  int greedy, attention;
  int start_token, end_token, pad_token;
  // memory
  const float* keys;      // [B,Tm,128]
  const float* values;    // [B,Tm,256]  (= enc_output; masked positions never contribute)
  const uint8_t* mask;    // [B,Tm]
  // recurrent state, N = B*W rows
  // StackedRNNCells (basecaller.py:85-91): layer k's buffers sit at base + k * ls_* elements
  int depth;              // decoder_depth
  size_t ls_xh, ls_c;     // layer strides (elements) of xh and of c / c_new / h_new
  float* xh;              // [depth][N,256]: layer 0 = [attention | h_0], layer k>=1 = [h_{k-1} (new) | h_k]
  float* z;               // [N,512] gate pre-activations (unused by the fused cell kernel)
  float* c;               // [depth][N,128]
  float* c_new;           // [depth][N,128]
  float* h_new;           // [depth][N,128]
  int* tok;               // [N]
  float* log_probs;       // [N]
  uint8_t* finished;      // [N]
  int* lengths;           // [N]
  // weights
  const float* W_att;     // [384,128]
  const float* W_fc;      // [128,V]
  const float* b_fc;      // [V]
  const float* W_q;       // [128,128]
  const float* v_att;     // [128]
  // per-step records, time-major
  int* step_ids;          // [L-1,B,W]
  int* parent_ids;        // [L-1,B,W]
  float* step_scores;     // [L-1,B,W]
  float* step_logits;     // [L-1,B,W,V]   (greedy or debug) or nullptr
  float* step_align;      // [L-1,B,W,Tm]  (debug) or nullptr
  int* nfin;              // [L] chunks-finished counter per step
  // optional fused post-processing of the best beam (basecaller.py:289-294 + utils.py:123-128)
  uint8_t* call_bases;    // [B,L-1] compacted base letters of tokens_to_nuc_sequences (or nullptr)
  float* call_probs;      // [B,L-1] exp(score_t - score_{t-1})
  int* call_len;          // [B] letters per chunk
  uint8_t lut[RV_MAX_VOCAB];   // token id -> upper-case letter, 0 for tokens the string form drops
  int* chunk_steps;       // persistent decode: [B] steps each chunk ran (nullptr on the per-step graph path)
  int* S_dev;             // [8]: [0] = S of the whole slab, [1+g] = S of sub-slab g
  int* S_host;            // device address of the host-mapped pinned word the call's S is left in (no copy launch for 4 bytes)
  // persistent decode, Luong, one cell: scores and context as split-f16 MFMAs.  Scales: powers of two that bring the largest
  // value a key / a U' element can take (from the weights; |enc_out| <= 1) into [2^13, 2^14); descale = 2^-14 / scale (the query
  // and the alignments are scaled by 2^14)
  int mx_attention; float mx_kscale, mx_kdescale, mx_uscale, mx_udescale;
  // ... and the cell product [ctx' | h] . Wcat2 too (mx_attention == 2): Wc16 = Wcat2 as MFMA B fragments of two f16 parts,
  // [8 waves][32 (k-step, gate) pairs][2 parts][64 lanes][8 f16]; wave w owns units 16 w .. 16 w + 15 of all four gates; lane (n, kq)
  // of pair p = 4 ks + g holds T . Wcat2[k][128 g + 16 w + n] / xs[k], k = 32 ks + 8 kq + 0..7, with xs = mx_uscale for the ctx' rows
  // (k < 128) and 2^14 for the h rows -- the factors the inputs' f16 images carry -- and T the power of two that brings the largest
  // such element into [2^13, 2^14); mx_cdescale = 1 / T
  const uint16_t* Wc16; float mx_cdescale;
  // ... and the output layer of that decode: Wl16 = [W_fc ; A_h W_fc] [256][16 (V padded)] as B fragments of two f16 parts,
  // [8 k-steps][2 parts][64 lanes][8 f16] (16 KB), rows divided like Wc16's, one power-of-two scale; mx_ldescale = its inverse.
  // Wave 0 takes the logits of all beams as 24 MFMAs on the [ctx' | h] image while the other waves already stream the cell product
  const uint16_t* Wl16; float mx_ldescale;
  // Bahdanau on the matrix pipe (mx_attention == 2 with attention == 1): W_q [128][128] as B fragments of two f16 parts,
  // [8 waves][4 k-steps][2 parts][64 lanes][8 f16] (64 KB): wave w owns columns 16 w .. 16 w + 15, lane (n, kq) of k-step ks holds
  // T . W_q[32 ks + 8 kq + 0..7][16 w + n]; the A operand is the h half of the [ctx' | h] image (h 2^14): mx_qdescale = 2^-14 / T
  const uint16_t* Wq16; float mx_qdescale;
  // Two decoder cells on the matrix pipe (depth == 2, mx_attention == 2).  Wc16 is then cell 0's product over [ctx' | h_1 | h_0]:
  // [W_a ; A_h W_a ; U_0] [384][512] as [8 waves][48 (k-step, gate) pairs][2 parts][64 lanes][8 f16] (768 KB; rows divided by mx_uscale /
  // 2^14 / 2^14, one power-of-two scale, mx_cdescale its inverse); W1c16 = cell 1's two products, [8 waves][32 pairs][2][64][8] (512 KB):
  // pairs 0..15 = W_1 (input product on h_0), pairs 16..31 = U_1 (recurrent product on h_1), rows divided by 2^14, one scale (mx_c1descale)
  const uint16_t* W1c16; float mx_c1descale;
  int attend_threads;     // 0: pick by slab size; 256 / 512: force that single-pass attend variant
  int part;               // sub-slab index (decode of one slab may run as up to 4 concurrent sub-slabs)
  long long* dbg_ts;      // diagnostic: [16] s_memtime stamps of block 0 at the phase boundaries of step 3
  int dbg_stop;           // diagnostic builds only: leave k_dec_attend after phase N (0 = run everything)
};
hipError_t configure_decode_kernels();   // per-device dynamic-LDS opt-in; call after hipSetDevice; first error or hipSuccess
void launch_dec_init(const DecState& d, hipStream_t s);
// layer: which stacked cell; Wtok (one-hot embedding rows) is non-null for layer 0 only
void launch_dec_cell(const DecState& d, int layer, const float* WcatT /*[512,256] = ([W_in;U])^T*/, const float* Wtok /*[V,512]*/,
                     const float* bias /*[512]*/, int step, hipStream_t s);
// flash: single-pass Luong attend over `values` only (WmemT = W_mem^T [128,256]); else the two-pass kernel
void launch_dec_attend(const DecState& d, const float* WmemT, bool flash, int step, hipStream_t s);
// Persistent decode (Luong beam search with W <= 8, W <= 5 with two stacked cells; greedy search; no taps): the whole loop in
// one launch, the chunk's attention memory resident in registers; also writes S_dev[0..1].  d.values must point at the
// PROJECTED memory [B,Tm,256] = enc_output . [W_mem | A_c] (keys | attention-layer image of the values).
// One cell: Wcat = [W_a ; U + A_h W_a] and Nh = A_h W_fc [128,V] (the attention layer's h part folded into the weights);
// two cells: Wcat = [W_a ; U_0], Wcat1 / bdec1 = second cell ([W_1;U_1], b_1), Nh unused.
bool dec_persist_supported(const DecState& d);
void launch_dec_persist(const DecState& d, const float* Wcat /*[256,512]*/, const float* Wtok /*[V,512]*/,
                        const float* bdec /*[512]*/, const float* Wcat1, const float* bdec1, const float* Nh, hipStream_t s);
// ptab != null: the two output addresses are read from ptab[RV_PTAB_TOKENS] / ptab[RV_PTAB_OUT2] on the device instead
void launch_dec_finalize(const DecState& d, int32_t* tokens /*[B,L-1]*/, float* scores_or_logits, hipStream_t s, const void* const* ptab = nullptr);
struct DecParts { const int* nfin[4]; int B[4]; int n; int steps; int* S_dev; int* S_host; };
void launch_dec_reduce_steps(const DecParts& p, hipStream_t s);   // S_dev[0] = max_g S_g, S_dev[1+g] = S_g

// ---------------------------------------------------------------- device math helpers
#ifdef __HIPCC__
__device__ __forceinline__ float rv_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float rv_tanh(float x) {
  // tanh(x) = 2*sigmoid(2x) - 1 ; absolute error ~1e-7, saturates cleanly
  return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)), -1.0f);
}
#endif
