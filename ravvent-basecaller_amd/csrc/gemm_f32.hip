// K0 / K2 -- exact-fp32 MFMA GEMM  C[M,N] = A[M,K] . B[K,N] (+ bias, + gathered row, * row mask)
// on v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate: bit-for-bit a k-ordered fmaf chain).
//
// Used for the dense contractions of the path only:
//   K0  layer>=1 input projection  [B*T,256] x [256,512] per direction  (Keras LSTMCell x.W + b,
//       SURVEY.md A.1; the reference does it per timestep inside tf.while_loop)
//   K2  attention keys             [B*T_m,256] x [256,128], rows of padded steps zeroed
//       (setup_memory: values = memory*mask, keys = values.W_mem -- /root/reference/basecaller.py:303)
//   D1  decoder cell pre-activations [B*W,256] x [256,512] + W[one_hot(token)] + b
//       (cell input = concat(one_hot, attention) -- SURVEY.md A.4)
//
// Block = 4 waves (2x2), each wave TMxTN tiles of 32x32; K step 16, double-buffered LDS with the
// next tile's global loads held in registers across the MFMA phase.  A is stored k-major in LDS
// (row pad 2 -> conflict-free ds_write_b32 of the transposed float4s, conflict-free operand reads).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int TM, int TN, int BK>
__global__ __launch_bounds__(256) void k_gemm_f32(GemmArgs a) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int KV = BK / 4;                 // float4 per A row per K tile
  constexpr int LDA_S = BM + 2, LDB_S = BN + 4;
  __shared__ __align__(16) float As[2][BK * LDA_S];
  __shared__ __align__(16) float Bs[2][BK * LDB_S];

  if (a.skip_flag && *a.skip_flag >= a.skip_when) return;
  // Tile coordinates.  Plain: (x = N tile, y = M tile, z = problem).  XCD-aware: workgroups are dealt
  // round-robin to the 8 XCDs (private L2s), so ids {x, x+8, x+16, ...} share an XCD; the
  // ntn*nprob workgroups that read the same A tile are made consecutive on ONE XCD -- the A
  // panel is then fetched from HBM once instead of once per XCD (PMC: 8.4x -> ~1x the bytes).
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.xcd_remap) {
    const int ntn = a.N / BN, nprob = a.Bm1 ? 2 : 1, per = ntn * nprob;
    const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
    const int combo = j % per;
    by = (j / per) * 8 + xcd;
    bx = combo % ntn; bz = combo / ntn;
    if (by * BM >= a.M) return;
  }
  if (bz == 1) { a.Bm = a.Bm1; a.bias = a.bias1; a.C = a.C1; }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = by * BM, n0 = bx * BN;
  const int nk = a.K / BK;

  constexpr int NA = TM * KV / 4, NB = TN * KV / 4;   // float4 loads per thread per K tile
  float4 ra[NA], rb[NB];
  auto gload = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      const int idx = tid + 256 * p, row = idx / KV, kv = idx % KV;
      const int m = m0 + row;
      ra[p] = m < a.M ? *reinterpret_cast<const float4*>(a.A + (size_t)m * a.lda + k0 + 4 * kv)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      const int idx = tid + 256 * p, kr = idx / (BN / 4), nv = idx % (BN / 4);
      rb[p] = *reinterpret_cast<const float4*>(a.Bm + (size_t)(k0 + kr) * a.ldb + n0 + 4 * nv);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      const int idx = tid + 256 * p, row = idx / KV, kv = idx % KV;
      float* d = &As[buf][(4 * kv) * LDA_S + row];
      d[0] = ra[p].x; d[LDA_S] = ra[p].y; d[2 * LDA_S] = ra[p].z; d[3 * LDA_S] = ra[p].w;
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      const int idx = tid + 256 * p, kr = idx / (BN / 4), nv = idx % (BN / 4);
      *reinterpret_cast<float4*>(&Bs[buf][kr * LDB_S + 4 * nv]) = rb[p];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const float* as = &As[buf][(lane >> 5) * LDA_S + wm * 32 * TM + (lane & 31)];
    const float* bs = &Bs[buf][(lane >> 5) * LDB_S + wn * 32 * TN + (lane & 31)];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = as[kk * LDA_S + 32 * i];
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) bv[jn] = bs[kk * LDB_S + 32 * jn];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[jn], acc[i][jn], 0, 0, 0);
    }
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 32 * TM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m >= a.M) continue;
      const bool keep = a.row_mask ? a.row_mask[m] != 0 : true;
      const float* grow = a.gather_idx ? a.gather_tab + (size_t)a.gather_idx[m] * a.ld_tab : nullptr;
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        const int n = n0 + wn * 32 * TN + 32 * jn + (lane & 31);
        float v = acc[i][jn][r];
        if (grow) v += grow[n];
        if (a.bias) v += a.bias[n];
        a.C[(size_t)m * a.ldc + n] = keep ? v : 0.f;
      }
    }
  }
}

}  // namespace

void launch_gemm_f32(const GemmArgs& a, bool small_tile, hipStream_t s) {
  if (small_tile) {
    dim3 grid(a.N / 64, (a.M + 63) / 64);
    hipLaunchKernelGGL((k_gemm_f32<1, 1, 16>), grid, dim3(256), 0, s, a);
  } else {
    const int ntm = (a.M + 127) / 128, ntn = a.N / 128, nprob = a.Bm1 ? 2 : 1;
    dim3 grid(ntn, ntm, nprob);
    if (a.xcd_remap) grid = dim3(((ntm + 7) / 8) * 8 * ntn * nprob, 1, 1);
    hipLaunchKernelGGL((k_gemm_f32<2, 2, 16>), grid, dim3(256), 0, s, a);
  }
}
