// MFMA tile helpers shared by the fused kernels (gfx950, wave64).
//
// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// the 16 result registers of a lane hold column l&31, rows (reg&3) + 8*(reg>>2) + 4*(l>>5).
// v_mfma_f32_16x16x4_f32: lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
// the 4 result registers hold column l&15, rows 4*(l>>4) + reg.
// Operands live in LDS "images": [row][feature] with an odd row stride, so that the operand
// reads of forward, input-gradient and weight-gradient products are all bank-conflict-free.
#pragma once
#include "common.h"

namespace hhrl {
namespace mt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int drow(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__device__ __forceinline__ void wave_sync() {
  // LDS traffic of one wave is processed in order; this only stops the
  // compiler from moving LDS accesses across the hand-off point.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ f32x16 splat(float v) {
  f32x16 r;
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = v;
  return r;
}

// acc[t] += A[32 x K] * B[K x 32] for TJ column tiles, operands in LDS:
//   A(i,k) = a[i*ASI + k*ASK]      B(k, 32*t + j) = b[k*BSK + (32*t + j)*BSJ]
template <int TJ, int K, int ASI, int ASK, int BSK, int BSJ>
__device__ __forceinline__ void wave_mma(f32x16 (&acc)[TJ], const float* a, const float* b, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const float* ap = a + r * ASI + h * ASK;
  const float* bp = b + h * BSK + r * BSJ;
#pragma unroll
  for (int k = 0; k < K; k += 2) {
    const float av = ap[k * ASK];
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
      const float bv = bp[k * BSK + t * 32 * BSJ];
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
    }
  }
}

// Store a D tile into an image: img[row][col0 + lane&31], rows = drow(reg, h).
template <int LD>
__device__ __forceinline__ void store_tile(float* img, int col0, const f32x16& t, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 16; ++i) img[drow(i, h) * LD + col0 + r] = t[i];
}

// Copy a [rows][cols] row-major global matrix into an LDS image with row stride ld (generic, scalar).
__device__ __forceinline__ void load_weight(float* img, int ld, const float* __restrict__ w, int rows, int cols, int tid, int nthreads) {
  for (int idx = tid; idx < rows * cols; idx += nthreads) {
    const int r = idx / cols, c = idx - r * cols;
    img[r * ld + c] = w[idx];
  }
}

// Weight matrices [ROWS][COLS] (row-major, 16-byte aligned, COLS % 32 == 0) travel as float4 per thread.
// Thread -> element map: a group of 32 threads takes a block of 4 rows x 32 columns, 8 threads per row.
// Global reads stay in 128-byte runs, and with an image row stride of 1 mod 32 the four 4-byte LDS
// writes of a float4 land in 32 different banks across the group.  NT = threads that share the copy.
template <int ROWS, int COLS, int NT = 256>
struct W4Map {
  static constexpr int GROUPS = NT / 32;
  static_assert(COLS % 32 == 0 && ROWS % 4 == 0 && ((ROWS / 4) * (COLS / 32)) % GROUPS == 0, "unsupported weight shape");
  static constexpr int CB = COLS / 32, STEPS = (ROWS / 4) * CB / GROUPS;
  __device__ __forceinline__ static void pos(int tid, int u, int& row, int& col) {
    const int blk = u * GROUPS + (tid >> 5), br = blk / CB, bc = blk % CB;
    row = 4 * br + ((tid & 31) >> 3);
    col = 32 * bc + 4 * (tid & 7);
  }
};
template <int ROWS, int COLS, int STEPS, int NT = 256>
__device__ __forceinline__ void request_w4(float4 (&v)[STEPS], const float* __restrict__ w, int tid) {
  static_assert(STEPS == W4Map<ROWS, COLS, NT>::STEPS, "register array does not match the weight shape");
#pragma unroll
  for (int u = 0; u < STEPS; ++u) {
    int row, col;
    W4Map<ROWS, COLS, NT>::pos(tid, u, row, col);
    v[u] = *reinterpret_cast<const float4*>(w + row * COLS + col);
  }
}
template <int ROWS, int COLS, int STEPS, int NT = 256>
__device__ __forceinline__ void commit_w4(float* img, int ld, const float4 (&v)[STEPS], int tid) {
  static_assert(STEPS == W4Map<ROWS, COLS, NT>::STEPS, "register array does not match the weight shape");
#pragma unroll
  for (int u = 0; u < STEPS; ++u) {
    int row, col;
    W4Map<ROWS, COLS, NT>::pos(tid, u, row, col);
    float* p = img + row * ld + col;
    p[0] = v[u].x; p[1] = v[u].y; p[2] = v[u].z; p[3] = v[u].w;
  }
}

// One 16x16 output block: init + A[16 x K] * B[K x 16], operands in LDS
//   A(i,k) = a[i*ASI + k*ASK]      B(k,j) = b[k*BSK + j*BSJ]
// Result: column lane&15, rows 4*(lane>>4) + reg.  Two accumulation chains (even / odd groups of four k).
template <int K, int ASI, int ASK, int BSK, int BSJ>
__device__ __forceinline__ f32x4 wave_mma16(const float* a, const float* b, int lane, f32x4 init) {
  static_assert(K % 8 == 0, "K must be a multiple of 8");
  const int i = lane & 15, q = lane >> 4;
  const float* ap = a + i * ASI + q * ASK;
  const float* bp = b + q * BSK + i * BSJ;
  f32x4 c0 = init, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < K; k += 8) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[k * ASK], bp[k * BSK], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[(k + 4) * ASK], bp[(k + 4) * BSK], c1, 0, 0, 0);
  }
  return c0 + c1;
}
// the same with one start value for the whole column (a bias)
template <int K, int ASI, int ASK, int BSK, int BSJ>
__device__ __forceinline__ f32x4 wave_mma16(const float* a, const float* b, int lane, float init) {
  const f32x4 v = {init, init, init, init};
  return wave_mma16<K, ASI, ASK, BSK, BSJ>(a, b, lane, v);
}
// img[(row0 + row)][col0 + col] = block
template <int LD>
__device__ __forceinline__ void store_block16(float* img, int row0, int col0, const f32x4& c, int lane) {
  const int j = lane & 15, q = lane >> 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) img[(row0 + 4 * q + e) * LD + col0 + j] = c[e];
}

}  // namespace mt
}  // namespace hhrl
