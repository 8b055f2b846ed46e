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

// Latency-oriented variant for small outputs (batch-64 agent updates, LSTM steps).
// One workgroup = one 32x32 output tile; its NW waves (4 or 8) each take a share of the reduction range,
// staging their operand chunks in wave-private LDS (no workgroup barrier in the loop), and the partial
// tiles are summed in LDS in a fixed order.  Many small workgroups instead of a few long-running ones.
// The staging is bound by instruction issue, so eight waves (two per SIMD, 32-element chunks) halve it.
template <int NW>
__device__ __forceinline__ void gemm_small_body(const GemmArgs& g, int bx, int by, int bz) {
  constexpr int SKC = NW == 4 ? 64 : 32;            // reduction elements staged per wave per chunk
  constexpr int RPL = SKC / 2;                      // operand elements per lane per chunk: 64 lanes x RPL = 32 x SKC
  constexpr int NT = 64 * NW;
  __shared__ float stage[NW][2][SKC][33];
  __shared__ float rsum[NW][32];
  // the partial output tile of a wave takes the place of its own staging buffers once its products are done
  // (67 KB of LDS per workgroup instead of 101 KB: two workgroups per CU)
  auto part = [&](int w) -> float (*)[33] { return stage[w][0]; };
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // in a scalar register: what depends on it is wave-uniform
  const int r = lane & 31, h = lane >> 5;
  const int kl = lane & (SKC - 1), hf = lane / SKC; // k-contiguous staging: lane -> (k of the chunk, half of the 32 rows)
  const int i0 = by * 32, j0 = bx * 32;
  const int zbeg = bz * g.k_chunk;
  const int zend = min(g.K, zbeg + g.k_chunk);
  // this wave's share of [zbeg, zend), a multiple of 2 except at the end
  const int span = (((zend - zbeg + NW - 1) / NW) + 1) & ~1;
  const int kbeg = min(zend, zbeg + wave * span), kend = min(zend, kbeg + span);
  float (*As)[33] = stage[wave][0];
  float (*Bs)[33] = stage[wave][1];

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  // Rows of A beyond M and columns of B beyond N only feed output elements that are never stored, so
  // their addresses are clamped to the last valid row / column instead of predicating every load, and
  // k-strided operands are read pair by pair with the pointer held at the last full pair.  Only the
  // reduction range needs exact values: the last element of an odd chunk and the zero row after it are
  // written separately.
  const int nra = min(32, g.M - i0), ncb = min(32, g.N - j0);
  // The chunks are software-pipelined: the loads of chunk n+1 are issued after chunk n has been written to
  // LDS and before its products, so one memory latency is exposed per wave instead of one per chunk.
  float av[RPL], bv[RPL], mv[RPL], a_tail = 0.f, b_tail = 0.f, m_tail = 0.f;
  const bool masked = g.Y != nullptr;
  const bool want_rs = g.rowsum != nullptr && bx == 0;
  float rs = 0.f;
  auto request = [&](int k0) {
    const int kc = min(SKC, kend - k0);
    const int np = kc >> 1;                             // full (k, k+1) pairs in the chunk
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
      av[i] = 0.f;
      bv[i] = 0.f;
      mv[i] = 0.f;
    }
    a_tail = 0.f;
    b_tail = 0.f;
    m_tail = 0.f;
    if (g.a_kcontig) {
      if (kl < kc) {
        const int first = hf * RPL;                     // this lane's rows: first .. first + RPL - 1
        const float* pa = g.A + (long)(i0 + min(first, nra - 1)) * g.a_si + (k0 + kl);
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
          av[i] = *pa;
          pa += (first + i + 1 < nra) ? g.a_si : 0;
        }
        if (masked) {
          const float* pm = g.Y + (long)(i0 + min(first, nra - 1)) * g.y_si + (long)(k0 + kl) * g.y_sk;
#pragma unroll
          for (int i = 0; i < RPL; ++i) {
            mv[i] = *pm;
            pm += (first + i + 1 < nra) ? g.y_si : 0;
          }
        }
      }
    } else {
      const float* row = g.A + (long)(i0 + min(r, nra - 1)) * g.a_si + (long)k0 * g.a_sk;
      if (np > 0) {
        const float* pa = row + h * g.a_sk;
#pragma unroll
        for (int t = 0; t < RPL; ++t) {
          av[t] = *pa;
          pa += (t + 1 < np) ? 2 * g.a_sk : 0;
        }
      }
      if (kc & 1) a_tail = row[(long)(kc - 1) * g.a_sk];
      if (masked) {
        const float* mrow = g.Y + (long)(i0 + min(r, nra - 1)) * g.y_si + (long)k0 * g.y_sk;
        if (np > 0) {
          const float* pm = mrow + h * g.y_sk;
#pragma unroll
          for (int t = 0; t < RPL; ++t) {
            mv[t] = *pm;
            pm += (t + 1 < np) ? 2 * g.y_sk : 0;
          }
        }
        if (kc & 1) m_tail = mrow[(long)(kc - 1) * g.y_sk];
      }
    }
    if (g.b_kcontig) {
      if (kl < kc) {
        const int first = hf * RPL;
        const float* pb = g.B + (long)(j0 + min(first, ncb - 1)) * g.b_sj + (k0 + kl);
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
          bv[j] = *pb;
          pb += (first + j + 1 < ncb) ? g.b_sj : 0;
        }
      }
    } else {
      const float* col = g.B + (long)k0 * g.b_sk + (long)(j0 + min(r, ncb - 1)) * g.b_sj;
      if (np > 0) {
        const float* pb = col + h * g.b_sk;
#pragma unroll
        for (int t = 0; t < RPL; ++t) {
          bv[t] = *pb;
          pb += (t + 1 < np) ? 2 * g.b_sk : 0;
        }
      }
      if (kc & 1) b_tail = col[(long)(kc - 1) * g.b_sk];
    }
  };
  // the epilogue's bias is requested with the first chunk
  float bias_v[1024 / NT];
#pragma unroll
  for (int u = 0; u < 1024 / NT; ++u) {
    const int gj = j0 + ((u * NT + tid) & 31);
    bias_v[u] = (g.bias && g.mode != 2 && gj < g.N) ? g.bias[gj] : 0.f;
  }
  if (kbeg < kend) request(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += SKC) {
    const int kc = min(SKC, kend - k0);
    const int kc2 = (kc + 1) & ~1;
    // As[k][i] = A(i0+i, k0+k), Bs[k][j] = B(k0+k, j0+j)
    if (masked) {                                       // one branch on the activation per chunk, not per element
      if (g.y_act == HHRL_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < RPL; ++i) av[i] = mv[i] > 0.f ? av[i] : 0.f;
        a_tail = m_tail > 0.f ? a_tail : 0.f;
      } else if (g.y_act == HHRL_ACT_TANH) {
#pragma unroll
        for (int i = 0; i < RPL; ++i) av[i] *= 1.f - mv[i] * mv[i];
        a_tail *= 1.f - m_tail * m_tail;
      } else if (g.y_act == HHRL_ACT_SIGMOID) {
#pragma unroll
        for (int i = 0; i < RPL; ++i) av[i] *= mv[i] * (1.f - mv[i]);
        a_tail *= m_tail * (1.f - m_tail);
      }
    }
    if (g.a_kcontig) {
#pragma unroll
      for (int i = 0; i < RPL; ++i) As[kl][hf * RPL + i] = av[i];
    } else {
#pragma unroll
      for (int t = 0; t < RPL; ++t) As[2 * t + h][r] = av[t];
      if (kc & 1) As[kc - 1 + h][r] = h == 0 ? a_tail : 0.f;
    }
    if (g.b_kcontig) {
#pragma unroll
      for (int j = 0; j < RPL; ++j) Bs[kl][hf * RPL + j] = bv[j];
    } else {
#pragma unroll
      for (int t = 0; t < RPL; ++t) Bs[2 * t + h][r] = bv[t];
      if (kc & 1) Bs[kc - 1 + h][r] = h == 0 ? b_tail : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (k0 + SKC < kend) request(k0 + SKC);
    if (want_rs) {
      for (int k = 0; k < kc2; k += 2) {
        const float a = As[k + h][r];
        rs += a;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bs[k + h][r], acc, 0, 0, 0);
      }
    } else {
      for (int k = 0; k < kc2; k += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k + h][r], Bs[k + h][r], acc, 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) part(wave)[(i & 3) + 8 * (i >> 2) + 4 * h][r] = acc[i];
  if (want_rs) {
    rs += __shfl_xor(rs, 32);                              // the two reduction parities of a row
    if (h == 0) rsum[wave][r] = rs;
  }
  __syncthreads();
  if (want_rs && tid < 32 && i0 + tid < g.M) {
    float v = (rsum[0][tid] + rsum[1][tid]) + (rsum[2][tid] + rsum[3][tid]);
    if (NW == 8) v += (rsum[4][tid] + rsum[5][tid]) + (rsum[6][tid] + rsum[7][tid]);
    unsafeAtomicAdd(&g.rowsum[i0 + tid], v);
    if (g.rowsum2) unsafeAtomicAdd(&g.rowsum2[i0 + tid], v);
  }
  // fixed-order sum of the partial tiles (pairs first), epilogue, store
#pragma unroll
  for (int u = 0; u < 1024 / NT; ++u) {
    const int idx = u * NT + tid;
    const int row = idx >> 5, col = idx & 31;
    const int gi = i0 + row, gj = j0 + col;
    if (gi >= g.M || gj >= g.N) continue;
    float v = (part(0)[row][col] + part(1)[row][col]) + (part(2)[row][col] + part(3)[row][col]);
    if (NW == 8) v += (part(4)[row][col] + part(5)[row][col]) + (part(6)[row][col] + part(7)[row][col]);
    v *= g.alpha;
    float* c = g.C + (long)gi * g.ldc + gj;
    if (g.mode == 2) {
      unsafeAtomicAdd(c, v);
    } else {
      if (g.bias) v += bias_v[u];
      if (g.mode == 1) v += *c;
      v = act_apply(v, g.act);
      *c = v;
      if (g.smooth_eps) {
        float n = 0.2f * g.smooth_eps[(long)gi * g.N + gj];
        n = fminf(fmaxf(n, -0.5f), 0.5f);
        g.y2[(long)gi * g.ldy2 + gj] = fminf(fmaxf(v + n, -1.f), 1.f);
      }
    }
  }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void gemm_small_kernel(GemmArgs g) { gemm_small_body<NW>(g, blockIdx.x, blockIdx.y, blockIdx.z); }

// Several independent small products in one launch: blockIdx.z = instance * splits + reduction slice.
// The grid is sized for the largest instance; workgroups outside an instance's tile range leave at once.
template <int NW>
__global__ __launch_bounds__(64 * NW) void gemm_small_group_kernel(GemmGroup grp) {
  const int inst = blockIdx.z / grp.splits, bz = blockIdx.z - inst * grp.splits;
  const GemmArgs& g = grp.g[inst];
  if ((int)blockIdx.x * 32 >= g.N || (int)blockIdx.y * 32 >= g.M || bz * g.k_chunk >= g.K) return;
  gemm_small_body<NW>(g, blockIdx.x, blockIdx.y, bz);
}


// The same for products with FEW output tiles (a batch-64 layer of 400 outputs is 26 tiles of 32 x 32: a tenth of
// the CUs, each with 8 x 32 matrix instructions of 64 cycles in a row): 16 x 16 output tiles on
// v_mfma_f32_16x16x4_f32, four times the workgroups, a quarter of the matrix-pipe time and half the operand bytes
// per workgroup.  Eight waves split the reduction range; operand chunks of 32 are staged in wave-private LDS
// ([k][row], stride 17), the partial tiles are summed in wave order.  Honours the same options as the 32 x 32 form.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gemm_tiny_body(const GemmArgs& g, int bx, int by, int bz) {
  constexpr int NW = 8, SKC = 32;
  __shared__ float stage[NW][2][SKC][17];
  __shared__ float rsum[NW][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = by * 16, j0 = bx * 16;
  const int zbeg = bz * g.k_chunk;
  const int zend = min(g.K, zbeg + g.k_chunk);
  const int span = (((zend - zbeg + NW - 1) / NW) + 3) & ~3;       // this wave's share, a multiple of 4 except at the end
  const int kbeg = min(zend, zbeg + wave * span), kend = min(zend, kbeg + span);
  float (*As)[17] = stage[wave][0];
  float (*Bs)[17] = stage[wave][1];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nra = min(16, g.M - i0), ncb = min(16, g.N - j0);
  // Lane roles while staging.  An operand that is contiguous along the reduction: lane = (k of the chunk, half of
  // the 16 rows), eight rows each; otherwise lane = (row, k mod 4), k = kq + 4 t.  Rows / columns beyond the matrix
  // are clamped to the last valid one (they feed outputs that are never stored); k beyond the range reads as zero.
  const int kl = lane & 31, hf = lane >> 5;
  const int rr = lane & 15, kq = lane >> 4;
  const bool masked = g.Y != nullptr;
  const bool want_rs = g.rowsum != nullptr && bx == 0;
  float rs = 0.f;
  // Requests carry no conditions: an element beyond the chunk's k range is asked for at the chunk's last valid k and
  // replaced by zero when the chunk is written to LDS.  (A conditional request is joined with its alternative where
  // it is issued, and that join waits for every request in flight: the chunk's requests then cost one memory round
  // trip each instead of one together.  The 32 x 32 kernel above keeps its conditional, pointer-stepping requests:
  // both unconditional forms tried there were slower, its joins fall where the values are needed anyway.)
  float av[8], bv[8], mv[8];
  auto request = [&](int k0) {
    const int kc = min(SKC, kend - k0);                  // >= 1
    // element t of this lane: (row, k) of A and of the mask Y
    long arow[8];
    int ak[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      arow[t] = g.a_kcontig ? i0 + min(8 * hf + t, nra - 1) : i0 + min(rr, nra - 1);
      ak[t] = k0 + (g.a_kcontig ? min(kl, kc - 1) : min(kq + 4 * t, kc - 1));
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) av[t] = g.A[arow[t] * g.a_si + (long)ak[t] * g.a_sk];
    if (masked) {
#pragma unroll
      for (int t = 0; t < 8; ++t) mv[t] = g.Y[arow[t] * g.y_si + (long)ak[t] * g.y_sk];
    }
    if (g.b_kcontig) {
      const int kk = k0 + min(kl, kc - 1);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const long col = j0 + min(8 * hf + t, ncb - 1);
        bv[t] = g.B[col * g.b_sj + kk];
      }
    } else {
      const long col = j0 + min(rr, ncb - 1);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int k = k0 + min(kq + 4 * t, kc - 1);
        bv[t] = g.B[(long)k * g.b_sk + col * g.b_sj];
      }
    }
  };
  float bias_v = 0.f;
  if (tid < 256 && g.bias && g.mode != 2 && j0 + (tid & 15) < g.N) bias_v = g.bias[j0 + (tid & 15)];
  if (kbeg < kend) request(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += SKC) {
    const int kc = min(SKC, kend - k0);
    const int kc4 = (kc + 3) & ~3;
    if (masked) {
      if (g.y_act == HHRL_ACT_RELU) {
#pragma unroll
        for (int t = 0; t < 8; ++t) av[t] = mv[t] > 0.f ? av[t] : 0.f;
      } else if (g.y_act == HHRL_ACT_TANH) {
#pragma unroll
        for (int t = 0; t < 8; ++t) av[t] *= 1.f - mv[t] * mv[t];
      } else if (g.y_act == HHRL_ACT_SIGMOID) {
#pragma unroll
        for (int t = 0; t < 8; ++t) av[t] *= mv[t] * (1.f - mv[t]);
      }
    }
    // every row of both images is written for every chunk (zeros beyond the chunk's k range)
    if (g.a_kcontig) {
      const bool in = kl < kc;
#pragma unroll
      for (int t = 0; t < 8; ++t) As[kl][8 * hf + t] = in ? av[t] : 0.f;
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) As[kq + 4 * t][rr] = kq + 4 * t < kc ? av[t] : 0.f;
    }
    if (g.b_kcontig) {
      const bool in = kl < kc;
#pragma unroll
      for (int t = 0; t < 8; ++t) Bs[kl][8 * hf + t] = in ? bv[t] : 0.f;
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) Bs[kq + 4 * t][rr] = kq + 4 * t < kc ? bv[t] : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (k0 + SKC < kend) request(k0 + SKC);
    for (int k = 0; k < kc4; k += 4) {
      const float a = As[k + kq][rr];
      if (want_rs) rs += a;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bs[k + kq][rr], acc, 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // result register e of lane (j = lane & 15, q = lane >> 4) is C(4q + e, j); the partial tile takes the place of
  // the wave's own A image
#pragma unroll
  for (int e = 0; e < 4; ++e) As[4 * kq + e][rr] = acc[e];
  if (want_rs) {
    rs += __shfl_xor(rs, 16);
    rs += __shfl_xor(rs, 32);
    if (lane < 16) rsum[wave][lane] = rs;
  }
  __syncthreads();
  if (want_rs && tid < 16 && i0 + tid < g.M) {
    const float v = ((rsum[0][tid] + rsum[1][tid]) + (rsum[2][tid] + rsum[3][tid])) +
                    ((rsum[4][tid] + rsum[5][tid]) + (rsum[6][tid] + rsum[7][tid]));
    unsafeAtomicAdd(&g.rowsum[i0 + tid], v);
    if (g.rowsum2) unsafeAtomicAdd(&g.rowsum2[i0 + tid], v);
  }
  if (tid < 256) {
    const int row = tid >> 4, col = tid & 15;
    const int gi = i0 + row, gj = j0 + col;
    if (gi < g.M && gj < g.N) {
      float v = ((stage[0][0][row][col] + stage[1][0][row][col]) + (stage[2][0][row][col] + stage[3][0][row][col])) +
                ((stage[4][0][row][col] + stage[5][0][row][col]) + (stage[6][0][row][col] + stage[7][0][row][col]));
      v *= g.alpha;
      float* c = g.C + (long)gi * g.ldc + gj;
      if (g.mode == 2) {
        unsafeAtomicAdd(c, v);
      } else {
        v += bias_v;
        if (g.mode == 1) v += *c;
        v = act_apply(v, g.act);
        *c = v;
        if (g.smooth_eps) {
          float n = 0.2f * g.smooth_eps[(long)gi * g.N + gj];
          n = fminf(fmaxf(n, -0.5f), 0.5f);
          g.y2[(long)gi * g.ldy2 + gj] = fminf(fmaxf(v + n, -1.f), 1.f);
        }
      }
    }
  }
}

__global__ __launch_bounds__(512) void gemm_tiny_kernel(GemmArgs g) { gemm_tiny_body(g, blockIdx.x, blockIdx.y, blockIdx.z); }
__global__ __launch_bounds__(512) void gemm_tiny_group_kernel(GemmGroup grp) {
  const int inst = blockIdx.z / grp.splits, bz = blockIdx.z - inst * grp.splits;
  const GemmArgs& g = grp.g[inst];
  if ((int)blockIdx.x * 16 >= g.N || (int)blockIdx.y * 16 >= g.M || bz * g.k_chunk >= g.K) return;
  gemm_tiny_body(g, blockIdx.x, blockIdx.y, bz);
}

// 16 x 16 tiles while the 32 x 32 tiles of the launch would leave three quarters of the CUs idle; HHRL_GEMM_TINY=0 turns them off.
static bool tiny_wanted(long tiles32) {
  static int limit = -1;
  if (limit < 0) {
    const char* e = getenv("HHRL_GEMM_TINY");       // 0: off; n > 1: use them up to n tiles of 32 x 32
    limit = e ? atoi(e) : 64;
    if (limit == 1) limit = 64;
  }
  return tiles32 <= limit;
}

// Waves per workgroup of the small kernels: 8 unless HHRL_GEMM_WAVES=4 asks for the four-wave form.
static int small_waves() {
  static int nw = 0;
  if (nw == 0) {
    const char* e = getenv("HHRL_GEMM_WAVES");
    nw = (e && atoi(e) == 4) ? 4 : 8;
  }
  return nw;
}

GemmArgs gemm_args(const float* A, long a_si, long a_sk, const float* B, long b_sk, long b_sj, float* C, long ldc,
                   int M, int N, int K, const float* bias, int act, int mode) {
  return GemmArgs{A, a_si, a_sk, B, b_sk, b_sj, C, ldc, bias, M, N, K, K, act, mode, a_sk == 1, b_sk == 1,
                  nullptr, 0, 0, HHRL_ACT_NONE, nullptr, nullptr, nullptr, nullptr, 0, 1.0f};
}

// Most 32 x 32 output tiles a product may have to be served by the small-product kernels (HHRL_GEMM_SMALL_MAX
// overrides).  Measured on HATD3Triple updates: 1.28 ms at batch 1024 and 4.1 ms at batch 4096 with them against
// 3.4 ms and 7.6 ms with the 64 x 64 tile kernel, so the latter is left to products beyond 8192 tiles.
static long small_limit() {
  static long lim = -1;
  if (lim < 0) {
    const char* e = getenv("HHRL_GEMM_SMALL_MAX");
    lim = e ? atol(e) : 8192;
  }
  return lim;
}
bool gemm_is_small(const GemmArgs& g) { return (long)ceil_div(g.M, 32) * ceil_div(g.N, 32) <= small_limit(); }

static int gemm_one(GemmArgs g, hipStream_t st);

// "<kernel>[n x MxNxK aK|aI bK|bJ mode m]" of a launch's first product while shapes are being profiled, else `kernel`
static const char* shape_name(const char* kernel, const GemmArgs& g, int n, int splits) {
  if (!prof_shapes()) return kernel;
  char buf[160];
  snprintf(buf, sizeof(buf), "%s[%dx%dx%dx%d,%s%s,mode%d,splits%d%s]", kernel, n, g.M, g.N, g.K, g.a_kcontig ? "aK" : "aI",
           g.b_kcontig ? "bK" : "bJ", g.mode, splits, g.Y ? ",mask" : "");
  return prof_intern(buf);
}

int gemm_group(const GemmArgs* gs, int n, hipStream_t st) {
  if (n <= 0) return HHRL_OK;
  // the grouped kernel serves latency-bound sizes; anything larger goes through the single-product path
  int max_m = 0, max_n = 0, max_k = 0;
  bool small = n <= kGemmGroupMax, reduce_mode = false;
  for (int i = 0; i < n; ++i) {
    const GemmArgs& g = gs[i];
    if (!gemm_is_small(g)) small = false;
    if (g.mode == 2) reduce_mode = true;
    max_m = g.M > max_m ? g.M : max_m;
    max_n = g.N > max_n ? g.N : max_n;
    max_k = g.K > max_k ? g.K : max_k;
  }
  if (!small || n == 1) {
    for (int i = 0; i < n; ++i) HHRL_TRY(gemm_one(gs[i], st));
    return HHRL_OK;
  }
  GemmGroup grp;
  grp.splits = 1;
  int chunk = 0;
  // Weight gradients reduce over the batch rows: from 1024 rows on the reduction is dealt over workgroups in slices of
  // 256 (float atomics combine them) -- a [300, 400] gradient is 130 tiles, half the CUs, each walking all 1024 rows of a
  // batch-1024 update otherwise (HATD3Triple at batch 1024: 1166 -> 1121 us per update; slices of 512: 1127, of 128: 1172).
  // HHRL_GEMM_SPLIT_CHUNK sets another slice length (sweeps).
  static int split_chunk = 0;
  if (split_chunk == 0) { const char* e = getenv("HHRL_GEMM_SPLIT_CHUNK"); split_chunk = (e && atoi(e) > 0) ? atoi(e) : 256; }
  if (reduce_mode && max_k >= 1024) {
    grp.splits = ceil_div(max_k, split_chunk) < 8 ? ceil_div(max_k, split_chunk) : 8;
    chunk = (ceil_div(max_k, grp.splits) + 7) & ~7;
    grp.splits = ceil_div(max_k, chunk);
  }
  double flops = 0.0, bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    grp.g[i] = gs[i];
    grp.g[i].k_chunk = (grp.splits > 1 && gs[i].mode == 2) ? chunk : gs[i].K;
    flops += 2.0 * gs[i].M * gs[i].N * gs[i].K;
    bytes += 4.0 * ((double)gs[i].M * gs[i].K + (double)gs[i].K * gs[i].N + (double)gs[i].M * gs[i].N);
  }
  long tiles32 = 0;
  for (int i = 0; i < n; ++i) tiles32 += (long)ceil_div(gs[i].M, 32) * ceil_div(gs[i].N, 32) * (gs[i].mode == 2 ? grp.splits : 1);
  if (tiny_wanted(tiles32)) {
    dim3 grid(ceil_div(max_n, 16), ceil_div(max_m, 16), n * grp.splits);
    ProfScope ps(shape_name("gemm_tiny_group_kernel", gs[0], n, grp.splits), st, flops, bytes);
    hipLaunchKernelGGL(gemm_tiny_group_kernel, grid, dim3(512), 0, st, grp);
    return launch_status("gemm_tiny_group_kernel");
  }
  dim3 grid(ceil_div(max_n, 32), ceil_div(max_m, 32), n * grp.splits);
  ProfScope ps(shape_name("gemm_small_group_kernel", gs[0], n, grp.splits), st, flops, bytes);
  if (small_waves() == 8) hipLaunchKernelGGL(gemm_small_group_kernel<8>, grid, dim3(512), 0, st, grp);
  else hipLaunchKernelGGL(gemm_small_group_kernel<4>, grid, dim3(256), 0, st, grp);
  return launch_status("gemm_small_group_kernel");
}

int gemm(const float* A, long a_si, long a_sk, const float* B, long b_sk, long b_sj, float* C, long ldc,
         int M, int N, int K, const float* bias, int act, int mode, hipStream_t st) {
  return gemm_one(gemm_args(A, a_si, a_sk, B, b_sk, b_sj, C, ldc, M, N, K, bias, act, mode), st);
}

static int gemm_one(GemmArgs g, hipStream_t st) {
  const float *A = g.A, *B = g.B;
  (void)A; (void)B;
  const int M = g.M, N = g.N, K = g.K, mode = g.mode;
  const long tiles32 = (long)ceil_div(M, 32) * ceil_div(N, 32);
  if (tiles32 <= small_limit()) {
    int gz = 1;
    static int split_chunk = 0;
    if (split_chunk == 0) { const char* e = getenv("HHRL_GEMM_SPLIT_CHUNK"); split_chunk = (e && atoi(e) > 0) ? atoi(e) : 256; }
    if (mode == 2 && K >= 1024) {
      // weight gradients over many rows: spread the reduction over workgroups too (atomics combine)
      long want = 1024 / tiles32;
      if (want < 1) want = 1;
      gz = (int)(ceil_div(K, split_chunk) < want ? ceil_div(K, split_chunk) : want);
      int chunk = ceil_div(K, gz);
      chunk = (chunk + 7) & ~7;
      g.k_chunk = chunk;
      gz = ceil_div(K, chunk);
    }
    if (tiny_wanted(tiles32 * gz)) {
      dim3 grid(ceil_div(N, 16), ceil_div(M, 16), gz);
      ProfScope ps(shape_name("gemm_tiny_kernel", g, 1, gz), st, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)K * N + (double)M * N));
      hipLaunchKernelGGL(gemm_tiny_kernel, grid, dim3(512), 0, st, g);
      return launch_status("gemm_tiny_kernel");
    }
    dim3 grid(ceil_div(N, 32), ceil_div(M, 32), gz);
    ProfScope ps(shape_name("gemm_small_kernel", g, 1, gz), st, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)K * N + (double)M * N));
    if (small_waves() == 8) hipLaunchKernelGGL(gemm_small_kernel<8>, grid, dim3(512), 0, st, g);
    else hipLaunchKernelGGL(gemm_small_kernel<4>, grid, dim3(256), 0, st, g);
    return launch_status("gemm_small_kernel");
  }
  HHRL_REQUIRE(!g.Y && !g.rowsum && !g.smooth_eps && g.alpha == 1.0f,
               "gemm: operand mask / row sums / smoothing / scaling are served by the small-product kernels only (M=%d N=%d)", M, N);
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
