// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), LDS-staged.
//
// One kernel serves the three products a dense layer needs, through strides:
//   forward   y  = x  * W^T   (both operands contiguous along the reduction)
//   input grad dx = dy * W     (B contiguous along the output column)
//   weight grad dW = dy^T * x  (reduction over the batch rows, split over
//                               workgroups and combined with float atomics)
// The f32 MFMA is an exact k-ordered fmaf chain, so results match a scalar
// fp32 implementation up to summation order.
//
// Tiling: 256 threads = 4 waves in a 2x2 grid; each wave owns TM x TN tiles of
// 32x32 (16 accumulator registers per tile).  The block tile is staged in LDS
// k-major ([BK][BM+1]) so that the 32 lanes of an MFMA operand read 32
// consecutive words: conflict-free ds_read_b32 in both 32-lane halves.
#include "common.h"

namespace hhrl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float* A;
  long a_si, a_sk;
  const float* B;
  long b_sk, b_sj;
  float* C;
  long ldc;
  const float* bias;
  int M, N, K;
  int k_chunk;
  int act, mode;
  int a_kcontig, b_kcontig;
};

constexpr int BK = 16;

template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(GemmArgs g) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  __shared__ float As[BK][BM + 1];
  __shared__ float Bs[BK][BN + 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i0 = blockIdx.y * BM, j0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * g.k_chunk;
  const int kend = min(g.K, kbeg + g.k_chunk);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
    for (int r = 0; r < BM * BK / 256; ++r) {
      const int idx = tid + r * 256;
      int kk, ii;
      if (g.a_kcontig) { kk = idx % BK; ii = idx / BK; } else { ii = idx % BM; kk = idx / BM; }
      const int gi = i0 + ii, gk = k0 + kk;
      float v = 0.f;
      if (gi < g.M && gk < kend) v = g.A[(long)gi * g.a_si + (long)gk * g.a_sk];
      As[kk][ii] = v;
    }
#pragma unroll
    for (int r = 0; r < BN * BK / 256; ++r) {
      const int idx = tid + r * 256;
      int kk, jj;
      if (g.b_kcontig) { kk = idx % BK; jj = idx / BK; } else { jj = idx % BN; kk = idx / BN; }
      const int gj = j0 + jj, gk = k0 + kk;
      float v = 0.f;
      if (gj < g.N && gk < kend) v = g.B[(long)gk * g.b_sk + (long)gj * g.b_sj];
      Bs[kk][jj] = v;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < BK; ks += 2) {
      float a[TM], b[TN];
      const int kk = ks + (lane >> 5);
#pragma unroll
      for (int t = 0; t < TM; ++t) a[t] = As[kk][(wm * TM + t) * 32 + (lane & 31)];
#pragma unroll
      for (int t = 0; t < TN; ++t) b[t] = Bs[kk][(wn * TN + t) * 32 + (lane & 31)];
#pragma unroll
      for (int ta = 0; ta < TM; ++ta)
#pragma unroll
        for (int tb = 0; tb < TN; ++tb)
          acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
    }
    __syncthreads();
  }

  // C/D layout of the 32x32 tile: column = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
#pragma unroll
  for (int ta = 0; ta < TM; ++ta) {
#pragma unroll
    for (int tb = 0; tb < TN; ++tb) {
      const int j = j0 + (wn * TN + tb) * 32 + (lane & 31);
      if (j >= g.N) continue;
      const float bj = g.bias ? g.bias[j] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + (wm * TM + ta) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (i >= g.M) continue;
        float* c = g.C + (long)i * g.ldc + j;
        float v = acc[ta][tb][r];
        if (g.mode == 2) {
          unsafeAtomicAdd(c, v);
        } else {
          v += bj;
          if (g.mode == 1) v += *c;
          *c = act_apply(v, g.act);
        }
      }
    }
  }
}

int gemm(const float* A, long a_si, long a_sk, const float* B, long b_sk, long b_sj, float* C, long ldc,
         int M, int N, int K, const float* bias, int act, int mode, hipStream_t st) {
  GemmArgs g{A, a_si, a_sk, B, b_sk, b_sj, C, ldc, bias, M, N, K, K, act, mode, a_sk == 1, b_sk == 1};
  const int TM = M > 1024 ? 2 : 1;
  const int TN = N > 64 ? 2 : 1;
  const int gx = ceil_div(N, 64 * TN), gy = ceil_div(M, 64 * TM);
  int gz = 1;
  if (mode == 2) {
    // Split the reduction so that ~1024 workgroups exist; each slice a multiple of BK.
    const int want = 1024 / (gx * gy) > 0 ? 1024 / (gx * gy) : 1;
    gz = ceil_div(K, 256) < want ? ceil_div(K, 256) : want;
    if (gz < 1) gz = 1;
    int chunk = ceil_div(K, gz);
    chunk = ceil_div(chunk, BK) * BK;
    g.k_chunk = chunk;
    gz = ceil_div(K, chunk);
  }
  dim3 grid(gx, gy, gz), block(256);
  static const char* const names[4] = {"gemm_mfma_kernel<1,1>", "gemm_mfma_kernel<1,2>", "gemm_mfma_kernel<2,1>", "gemm_mfma_kernel<2,2>"};
  ProfScope ps(names[(TM - 1) * 2 + (TN - 1)], st, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)K * N + (double)M * N));
  if (TM == 2 && TN == 2) hipLaunchKernelGGL((gemm_mfma_kernel<2, 2>), grid, block, 0, st, g);
  else if (TM == 2) hipLaunchKernelGGL((gemm_mfma_kernel<2, 1>), grid, block, 0, st, g);
  else if (TN == 2) hipLaunchKernelGGL((gemm_mfma_kernel<1, 2>), grid, block, 0, st, g);
  else hipLaunchKernelGGL((gemm_mfma_kernel<1, 1>), grid, block, 0, st, g);
  return launch_status("gemm_mfma_kernel");
}

}  // namespace hhrl
