// Three dense layers (Linear+ReLU, Linear+ReLU, Linear+act) as ONE kernel per direction, for the
// 128-wide networks of the worker agents (WorkerActor / WorkerCritic, algorithm.py:1187-1258, hidden
// sizes from configs.py:159): a batch-64 update is bound by the number of launches, and a network's
// 19 k parameters fit the registers of one workgroup.
//
// Weights live in REGISTERS, activations in LDS.  A workgroup is eight waves; wave w owns output columns
// 16w..16w+15 of both hidden layers and keeps its slices of W1, W2 (and its reduction range of W3) in
// registers for the whole launch, loaded straight from global memory: no LDS image of the weights, no
// staging phase, and a weight value is read once per workgroup however many rows the workgroup walks.
// Rows are processed in tiles of 16 (v_mfma_f32_16x16x4_f32, one 16x16 output block per wave and layer), so
// a 64-row minibatch spreads over four workgroups per network, and a 4096-row one over as many workgroups
// as fill the chip once, each looping over its tiles with the next tile's inputs in flight.
//
// Operand order of a product: the four values a lane holds in one 16-byte register group are four
// consecutive reduction indices 16m + 4q + e (q = lane/16); instruction e of group m therefore reduces over
// {16m + 4q + e : q = 0..3} -- any assignment works as long as both operands use the same one -- and both the
// activation rows (LDS, 16-byte reads) and the weight rows (global, 16-byte loads) are read along their
// contiguous dimension.
//
// The backward kernel keeps the transposed-use slices (W3, W2 by input column; W1 for the input gradient)
// and the weight-gradient accumulators of its column block in registers across all tiles of the
// workgroup.  At the end a workgroup hands its share over in one of two ways: with few workgroups per
// network (small batches) by float atomics into the gradient buffers; with many, as one row of a
// caller-provided slab [workgroups][parameters] that mlp3_slab_reduce_kernel sums in workgroup order and
// adds to the gradient buffers -- 256 workgroups adding 19 k floats each through atomics took four times
// as long as the whole rest of the kernel, and the slab sum is the same bits on every run.
#include "common.h"
#include "mfma_tiles.h"

namespace hhrl {
namespace mf {

using namespace mt;

constexpr int H = 128;          // both hidden widths
constexpr int KMAX = 64;        // widest input
constexpr int NMAX = 32;        // widest output
constexpr int NT = 512;         // eight waves
constexpr int ROWS = 16;        // rows per tile
constexpr int kMaxNets = HHRL_MLP3_GROUP_MAX;

struct Net {
  const float* x;
  const float *w1, *b1, *w2, *b2, *w3, *b3;
  float *h1, *h2, *y;
  const float* dy;
  float *gw1, *gb1, *gw2, *gb2, *gw3, *gb3, *dx;
  const float* smooth_eps;
  float* y2;
  float* slab;            // backward, optional: [gridDim.x][slab_floats(K0, N3)] partial parameter gradients instead of atomics
  float dy_const;
  int ldx, ldy, lddy, lddx, K0, N3, act3, ldy2, forward_dx;
  // backward, optional (tq1 != nullptr; N3 == 1): dL/dy is formed here from the TD3 target, dy = 2 inv_n (y - target) with
  // target = tr[row * ldr] + gamma * min(tq1[row], tq2[row]); inv_n * sum (y - target)^2 is added to *tloss, the
  // target stored to ty (optional)
  const float *tq1, *tq2, *tr;
  float *tloss, *ty;
  int ldr;
  float gamma, inv_n;
};
// order of a network's parameter gradients inside a slab row
__host__ __device__ inline int slab_w1(int, int) { return 0; }
__host__ __device__ inline int slab_b1(int K0, int) { return H * K0; }
__host__ __device__ inline int slab_w2(int K0, int) { return H * K0 + H; }
__host__ __device__ inline int slab_b2(int K0, int) { return H * K0 + H + H * H; }
__host__ __device__ inline int slab_w3(int K0, int) { return H * K0 + 2 * H + H * H; }
__host__ __device__ inline int slab_b3(int K0, int N3) { return H * K0 + 2 * H + H * H + N3 * H; }
// a slab row ends with one more entry, the workgroup's share of the loss (read only when the network forms one)
__host__ __device__ inline int slab_loss(int K0, int N3) { return H * K0 + 2 * H + H * H + N3 * H + N3; }
// (row length: a multiple of four floats, so that rows can be read 16 bytes at a time)
__host__ __device__ inline int slab_floats(int K0, int N3) { return (slab_loss(K0, N3) + 1 + 3) & ~3; }
struct NetGroup { Net in[kMaxNets]; };

// LDS map (floats).  Row strides are 4 mod 32 banks and a multiple of four floats: 16-byte reads of a row
// by the 16 lanes of a quarter wave, and the 4-byte writes of an output block, spread over all banks.
constexpr int LDA = H + 4;                       // hidden activation tiles [16][132]
constexpr int LDX = KMAX + 4;                    // input tile [16][68], columns >= K0 zero
constexpr int LDD = NMAX + 1;                    // gradient at layer 3's pre-activation [16][33], columns >= N3 zero
constexpr int LDP = 17;                          // partial 16x16 blocks
constexpr int NBX = KMAX / 16;                   // most 16-column blocks of the input
constexpr int L_X = 0;
constexpr int L_A1 = L_X + ROWS * LDX;
constexpr int L_A2 = L_A1 + ROWS * LDA;
constexpr int L_D3 = L_A2 + ROWS * LDA;
constexpr int L_P = L_D3 + ROWS * LDD;           // [8 waves][<= 4 blocks][16][17]: products split over the waves by reduction range
constexpr int L_END = L_P + 8 * NBX * 16 * LDP;
static_assert(L_END * 4 <= 65536, "the three-layer kernels use static-size LDS");

// Diagnostic builds (-DHHRL_STAMPS): shader-clock stamps of workgroup (0, 0), thread 0; hhrl_debug_mlp_stamps() copies them out.
#ifdef HHRL_STAMPS
__device__ long long g_mlp_stamps[2][64];
#define MLP_STAMP(k, i) do { const int _s = (i); if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && _s < 64) g_mlp_stamps[k][_s] = clock64(); } while (0)
#else
#define MLP_STAMP(k, i) do { } while (0)
#endif

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f32x4 mma4(f32x4 c, const float4& a, const float4& b) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
  return c;
}
__device__ __forceinline__ float4 lds4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---- weights of one wave, forward use: y[row][col0 + j] = sum_k a[row][k] * W[col0 + j][k]
struct WFwd {
  float4 w1[NBX];      // W1[col0 + j][16m + 4q ..]
  float4 w2[H / 16];   // W2[col0 + j][16m + 4q ..]
  float4 w3[NMAX / 16];// W3[16c + j][16w + 4q ..]: this wave's share of layer 3's reduction
  float b1, b2;
};
__device__ __forceinline__ void load_w1_fwd(WFwd& W, const Net& g, int col0, int j, int q) {
  const float* r = g.w1 + (long)(col0 + j) * g.K0;
#pragma unroll
  for (int m = 0; m < NBX; ++m) {
    const int k = 16 * m + 4 * q;
    W.w1[m] = make_float4(k < g.K0 ? r[k] : 0.f, k + 1 < g.K0 ? r[k + 1] : 0.f, k + 2 < g.K0 ? r[k + 2] : 0.f,
                          k + 3 < g.K0 ? r[k + 3] : 0.f);
  }
  W.b1 = g.b1[col0 + j];
}
__device__ __forceinline__ void load_w23_fwd(WFwd& W, const Net& g, int wave, int col0, int j, int q) {
#pragma unroll
  for (int m = 0; m < H / 16; ++m) W.w2[m] = *reinterpret_cast<const float4*>(g.w2 + (col0 + j) * H + 16 * m + 4 * q);
  W.b2 = g.b2[col0 + j];
#pragma unroll
  for (int c = 0; c < NMAX / 16; ++c)
    W.w3[c] = 16 * c + j < g.N3 ? *reinterpret_cast<const float4*>(g.w3 + (16 * c + j) * H + 16 * wave + 4 * q)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- weights of one wave, backward use: dx[row][col0 + j] = sum_o d[row][o] * W[o][col0 + j]
struct WBwd {
  float4 w3t[NMAX / 16];   // W3[16m + 4q + e][col0 + j]
  float4 w2t[H / 16];      // W2[16m + 4q + e][col0 + j]
  float4 w1t[NBX];         // W1[16w + 4q + e][16c + j]: this wave's share of the input gradient's reduction
};
__device__ __forceinline__ void load_w_bwd(WBwd& W, const Net& g, int wave, int col0, int j, int q, bool want_dx) {
#pragma unroll
  for (int m = 0; m < NMAX / 16; ++m) {
    const int n = 16 * m + 4 * q;
    const float* p = g.w3 + (long)n * H + col0 + j;
    W.w3t[m] = make_float4(n < g.N3 ? p[0] : 0.f, n + 1 < g.N3 ? p[H] : 0.f, n + 2 < g.N3 ? p[2 * H] : 0.f,
                           n + 3 < g.N3 ? p[3 * H] : 0.f);
  }
#pragma unroll
  for (int m = 0; m < H / 16; ++m) {
    const float* p = g.w2 + (16 * m + 4 * q) * H + col0 + j;
    W.w2t[m] = make_float4(p[0], p[H], p[2 * H], p[3 * H]);
  }
#pragma unroll
  for (int c = 0; c < NBX; ++c) {
    const int k = 16 * c + j;
    const float* p = g.w1 + (long)(16 * wave + 4 * q) * g.K0 + k;
    const bool ok = want_dx && k < g.K0;
    W.w1t[c] = ok ? make_float4(p[0], p[g.K0], p[2 * g.K0], p[3 * g.K0]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// ---- one tile's inputs on their way from global memory to LDS (requested one tile ahead)
struct TileIn {
  float x[ROWS * KMAX / NT];      // 2
  float4 h1, h2;                  // backward only
  float d3, y3;                   // backward only: dL/dy and the output y, combined by commit_bwd
  float t1, t2, tr;               // backward only: the target critic's two values and the reward of the row (TD3 target)
  bool okd;
};
__device__ __forceinline__ void request_x(TileIn& T, const Net& g, int row0, int M, int tid, int K0p) {
#pragma unroll
  for (int u = 0; u < ROWS * KMAX / NT; ++u) {
    const int e = u * NT + tid, rr = e / K0p, c = e - rr * K0p;
    T.x[u] = (rr < ROWS && c < g.K0 && row0 + rr < M) ? g.x[(long)(row0 + rr) * g.ldx + c] : 0.f;
  }
}
__device__ __forceinline__ void commit_x(float* lds, const TileIn& T, int tid, int K0p) {
#pragma unroll
  for (int u = 0; u < ROWS * KMAX / NT; ++u) {
    const int e = u * NT + tid, rr = e / K0p, c = e - rr * K0p;
    if (rr < ROWS) lds[L_X + rr * LDX + c] = T.x[u];
  }
}
__device__ __forceinline__ void request_bwd(TileIn& T, const Net& g, int row0, int M, int tid, int N3p) {
  const int rr = tid >> 5, c4 = (tid & 31) * 4;
  const bool ok = row0 + rr < M;
  T.h1 = ok ? *reinterpret_cast<const float4*>(g.h1 + (long)(row0 + rr) * H + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  T.h2 = ok ? *reinterpret_cast<const float4*>(g.h2 + (long)(row0 + rr) * H + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  // dL/dy and the output it is scaled by are only asked for here and combined when they are stored: combined here,
  // the values would be waited for before the requests that follow (the tile's inputs, the weights) are issued.
  // Threads without an element, and the case without a dy / without an output activation, ask for a word that
  // exists (the first of h1) and do not use it: no branch around a request, and no generic-address load.
  const int dr = tid / N3p, dc = tid - dr * N3p;
  const bool okd = dr < ROWS && dc < g.N3 && row0 + dr < M;
  const bool tgt = g.tq1 != nullptr, use_y = g.act3 != HHRL_ACT_NONE || tgt;
  const float* dyp = g.dy ? g.dy : g.h1;
  const float* yp = use_y ? g.y : g.h1;
  T.d3 = dyp[okd && g.dy ? (long)(row0 + dr) * g.lddy + dc : 0];
  T.y3 = yp[okd && use_y ? (long)(row0 + dr) * g.ldy + dc : 0];
  const long ti = okd && tgt ? row0 + dr : 0;
  T.t1 = (tgt ? g.tq1 : g.h1)[ti];
  T.t2 = (tgt ? g.tq2 : g.h1)[ti];
  T.tr = (tgt ? g.tr : g.h1)[ti * (tgt ? g.ldr : 0)];
  T.okd = okd;
}
// sq: this thread's sum of squared differences to the TD3 target so far (networks that form their own dL/dy)
__device__ __forceinline__ void commit_bwd(float* lds, const TileIn& T, const Net& g, int tid, int N3p, int row0, float& sq) {
  const int rr = tid >> 5, c4 = (tid & 31) * 4;
  *reinterpret_cast<float4*>(lds + L_A1 + rr * LDA + c4) = T.h1;
  *reinterpret_cast<float4*>(lds + L_A2 + rr * LDA + c4) = T.h2;
  const int dr = tid / N3p, dc = tid - dr * N3p;
  float v = 0.f;
  if (T.okd) {
    if (g.tq1 != nullptr) {                 // same arithmetic as td3_target_mse_kernel
      const float yi = T.tr + g.gamma * fminf(T.t1, T.t2);
      const float d = T.y3 - yi;
      sq += d * d;
      v = 2.f * g.inv_n * d;
      if (g.ty != nullptr) g.ty[row0 + dr] = yi;
    } else {
      v = g.dy ? T.d3 : g.dy_const;
    }
    if (g.act3 != HHRL_ACT_NONE) v *= act_grad_from_output(T.y3, g.act3);
  }
  if (dr < ROWS) lds[L_D3 + dr * LDD + dc] = v;
}

// Sum of the eight waves' partial blocks, in wave order: out(row, col) for col < 16 * nb
__device__ __forceinline__ float sum_partials(const float* lds, int nbmax, int row, int col) {
  const float* p = lds + L_P + (((col >> 4) * 16) + row) * LDP + (col & 15);
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) s += p[w * nbmax * 16 * LDP];
  return s;
}

// Weight-gradient accumulators of one wave's column block (rows of dW = this wave's output columns)
struct GradAcc {
  f32x4 w3[NMAX / 16];   // dW3[16c + 4q + e][col0 + j]
  f32x4 w2[H / 16];      // dW2[col0 + 4q + e][16c + j]
  f32x4 w1[NBX];         // dW1[col0 + 4q + e][16c + j]
  float b1, b2, b3;      // thread `tid`'s bias-gradient column (tid < 128 / N3)
};

// Backward of one tile that lies in LDS: X, A1 = h1, A2 = h2, D3.  pg: parameter gradients into `G`;
// want_dx: the input gradient to global memory.  Leaves the tile's LDS free (ends with a barrier).
__device__ __forceinline__ void tile_backward(float* lds, const Net& g, const WBwd& W, GradAcc& G, bool pg, bool want_dx, int row0, int M,
                                              int tid, int nm1, int nb3) {
  const int lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4, col0 = 16 * wave;
  float* X = lds + L_X;
  float* A1 = lds + L_A1;
  float* A2 = lds + L_A2;
  float* D3 = lds + L_D3;
  // ---- layer 3: dW3 += D3^T h2 (own columns of h2), db3, dh2 = (D3 W3) o relu'(h2) into the own columns of A2
  if (pg) {
    float b[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) b[e] = A2[(4 * q + e) * LDA + col0 + j];
#pragma unroll
    for (int c = 0; c < NMAX / 16; ++c)
      if (c < nb3) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          G.w3[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(D3[(4 * q + e) * LDD + 16 * c + j], b[e], G.w3[c], 0, 0, 0);
      }
    if (tid < g.N3) {
      float s = 0.f;
#pragma unroll
      for (int rr = 0; rr < ROWS; ++rr) s += D3[rr * LDD + tid];
      G.b3 += s;
    }
  }
  {
    f32x4 d = zero4();
#pragma unroll
    for (int m = 0; m < NMAX / 16; ++m)
      if (m < nb3) {
        const float* p = D3 + j * LDD + 16 * m + 4 * q;
        d = mma4(d, make_float4(p[0], p[1], p[2], p[3]), W.w3t[m]);
      }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float* p = A2 + (4 * q + e) * LDA + col0 + j;
      *p = *p > 0.f ? d[e] : 0.f;
    }
  }
  lds_barrier();
  // ---- layer 2: dW2 += dh2^T h1, db2, dh1 = (dh2 W2) o relu'(h1) into the own columns of A1
  {
    if (pg) {
      float a[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = A2[(4 * q + e) * LDA + col0 + j];
#pragma unroll
      for (int c = 0; c < H / 16; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          G.w2[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], A1[(4 * q + e) * LDA + 16 * c + j], G.w2[c], 0, 0, 0);
      if (tid < H) {
        float s = 0.f;
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) s += A2[rr * LDA + tid];
        G.b2 += s;
      }
    }
    f32x4 d = zero4();
#pragma unroll
    for (int m = 0; m < H / 16; ++m) d = mma4(d, lds4(A2 + j * LDA + 16 * m + 4 * q), W.w2t[m]);
    lds_barrier();                                     // every wave is done reading h1 and dh2
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float* p = A1 + (4 * q + e) * LDA + col0 + j;
      *p = *p > 0.f ? d[e] : 0.f;
    }
  }
  lds_barrier();
  // ---- layer 1: dW1 += dh1^T x, db1, dx = dh1 W1 (reduction split over the waves, summed in wave order)
  if (pg) {
    float a[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = A1[(4 * q + e) * LDA + col0 + j];
#pragma unroll
    for (int c = 0; c < NBX; ++c)
      if (c < nm1) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          G.w1[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], X[(4 * q + e) * LDX + 16 * c + j], G.w1[c], 0, 0, 0);
      }
    if (tid < H) {
      float s = 0.f;
#pragma unroll
      for (int rr = 0; rr < ROWS; ++rr) s += A1[rr * LDA + tid];
      G.b1 += s;
    }
  }
  if (want_dx) {
    const float4 a = lds4(A1 + j * LDA + col0 + 4 * q);
#pragma unroll
    for (int c = 0; c < NBX; ++c)
      if (c < nm1) {
        const f32x4 p = mma4(zero4(), a, W.w1t[c]);
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[L_P + ((wave * NBX + c) * 16 + 4 * q + e) * LDP + j] = p[e];
      }
    lds_barrier();
    for (int o = tid; o < ROWS * 16 * nm1; o += NT) {
      const int rr = o / (16 * nm1), col = o - rr * (16 * nm1);
      const float s = sum_partials(lds, NBX, rr, col);
      if (col < g.K0 && row0 + rr < M) g.dx[(long)(row0 + rr) * g.lddx + col] = s;
    }
  }
  lds_barrier();
}

// ------------------------------------------------------------------ forward
// DX: some network of the group also wants the input gradient for a constant dL/dy (forward_dx); the plain
// kernel does not carry the registers that takes.
template <bool DX>
__global__ __launch_bounds__(NT, 1) void mlp3_forward_kernel(NetGroup grp, int M, int tiles) {
  __shared__ __attribute__((aligned(16))) float lds[L_END];
  const Net& g = grp.in[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4, col0 = 16 * wave;
  const int K0p = (g.K0 + 15) & ~15, nm1 = K0p >> 4, N3p = (g.N3 + 15) & ~15, nb3 = N3p >> 4;
  const bool fdx = DX && g.forward_dx;
  int tile = blockIdx.x;
  if (tile >= tiles) return;
  MLP_STAMP(0, 0);
  int sn = 1;
  (void)sn;
  TileIn T;
  request_x(T, g, tile * ROWS, M, tid, K0p);
  WFwd W;
  load_w1_fwd(W, g, col0, j, q);
  load_w23_fwd(W, g, wave, col0, j, q);
  WBwd WB;
  if (DX) load_w_bwd(WB, g, wave, col0, j, q, fdx);
  float* X = lds + L_X;
  float* A1 = lds + L_A1;
  float* A2 = lds + L_A2;
  for (; tile < tiles; tile += gridDim.x) {
    const int row0 = tile * ROWS;
    commit_x(lds, T, tid, K0p);
    lds_barrier();
    MLP_STAMP(0, sn++);                                    // inputs (first tile: and the weights of layer 1) have arrived
    if (tile + (int)gridDim.x < tiles) request_x(T, g, (tile + gridDim.x) * ROWS, M, tid, K0p);
    // ---- layer 1 (K0 -> 128, ReLU)
    {
      f32x4 c = {W.b1, W.b1, W.b1, W.b1};
#pragma unroll
      for (int m = 0; m < NBX; ++m)
        if (m < nm1) c = mma4(c, lds4(X + j * LDX + 16 * m + 4 * q), W.w1[m]);
#pragma unroll
      for (int e = 0; e < 4; ++e) A1[(4 * q + e) * LDA + col0 + j] = fmaxf(c[e], 0.f);
    }
    lds_barrier();
    MLP_STAMP(0, sn++);
    // ---- layer 2 (128 -> 128, ReLU)
    {
      f32x4 c = {W.b2, W.b2, W.b2, W.b2};
#pragma unroll
      for (int m = 0; m < H / 16; ++m) c = mma4(c, lds4(A1 + j * LDA + 16 * m + 4 * q), W.w2[m]);
#pragma unroll
      for (int e = 0; e < 4; ++e) A2[(4 * q + e) * LDA + col0 + j] = fmaxf(c[e], 0.f);
    }
    lds_barrier();
    MLP_STAMP(0, sn++);
    // ---- layer 3 (128 -> N3 <= 32, act3): every wave reduces over its 16 hidden units, the partial blocks meet in LDS
    {
      const float4 a = lds4(A2 + j * LDA + col0 + 4 * q);
#pragma unroll
      for (int c = 0; c < NMAX / 16; ++c)
        if (c < nb3) {
          const f32x4 p = mma4(zero4(), a, W.w3[c]);
#pragma unroll
          for (int e = 0; e < 4; ++e) lds[L_P + ((wave * (NMAX / 16) + c) * 16 + 4 * q + e) * LDP + j] = p[e];
        }
    }
    if (g.h1 != nullptr && !fdx) {                         // the hidden activations for the backward call, 16 bytes per thread
      const int rr = tid >> 5, c4 = (tid & 31) * 4;
      if (row0 + rr < M) {
        *reinterpret_cast<float4*>(g.h1 + (long)(row0 + rr) * H + c4) = lds4(A1 + rr * LDA + c4);
        *reinterpret_cast<float4*>(g.h2 + (long)(row0 + rr) * H + c4) = lds4(A2 + rr * LDA + c4);
      }
    }
    lds_barrier();
    MLP_STAMP(0, sn++);
    for (int o = tid; o < ROWS * N3p; o += NT) {
      const int rr = o / N3p, col = o - rr * N3p, row = row0 + rr;
      float d3 = 0.f;
      if (col < g.N3 && row < M) {
        const float v = act_apply(sum_partials(lds, NMAX / 16, rr, col) + g.b3[col], g.act3);
        g.y[(long)row * g.ldy + col] = v;
        if (g.smooth_eps != nullptr) {      // target-policy smoothing of the same value, as td3_smooth_action_kernel
          float n = 0.2f * g.smooth_eps[(long)row * g.N3 + col];
          n = fminf(fmaxf(n, -0.5f), 0.5f);
          g.y2[(long)row * g.ldy2 + col] = fminf(fmaxf(v + n, -1.f), 1.f);
        }
        d3 = g.dy_const;
        if (g.act3 != HHRL_ACT_NONE) d3 *= act_grad_from_output(v, g.act3);
      }
      if (fdx) lds[L_D3 + rr * LDD + col] = d3;
    }
    lds_barrier();
    MLP_STAMP(0, sn++);
    if (DX) {
      if (fdx) {
        // the input gradient for dL/dy = dy_const on every row (the policy step's d(-mean Q)/d(action)): the tile's
        // activations are in LDS already; the hidden activations are overwritten by their gradients
        GradAcc none;
        tile_backward(lds, g, WB, none, false, true, row0, M, tid, nm1, nb3);
      }
    }
  }
}

// ------------------------------------------------------------------ backward
__global__ __launch_bounds__(NT, 1) void mlp3_backward_kernel(NetGroup grp, int M, int tiles) {
  __shared__ __attribute__((aligned(16))) float lds[L_END];
  const Net& g = grp.in[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4, col0 = 16 * wave;
  const int K0 = g.K0, K0p = (K0 + 15) & ~15, nm1 = K0p >> 4, N3 = g.N3, N3p = (N3 + 15) & ~15, nb3 = N3p >> 4;
  const bool pg = g.gw1 != nullptr;            // parameter gradients wanted
  const bool want_dx = g.dx != nullptr;
  int tile = blockIdx.x;
  if (tile >= tiles) return;
  MLP_STAMP(1, 0);
  int sn = 1;
  (void)sn;
  TileIn T;
  request_bwd(T, g, tile * ROWS, M, tid, N3p);
  request_x(T, g, tile * ROWS, M, tid, K0p);
  WBwd W;
  load_w_bwd(W, g, wave, col0, j, q, want_dx);
  GradAcc G;
#pragma unroll
  for (int c = 0; c < NMAX / 16; ++c) G.w3[c] = zero4();
#pragma unroll
  for (int c = 0; c < H / 16; ++c) G.w2[c] = zero4();
#pragma unroll
  for (int c = 0; c < NBX; ++c) G.w1[c] = zero4();
  G.b1 = G.b2 = G.b3 = 0.f;
  float sq = 0.f;
  for (; tile < tiles; tile += gridDim.x) {
    commit_x(lds, T, tid, K0p);
    commit_bwd(lds, T, g, tid, N3p, tile * ROWS, sq);
    lds_barrier();
    MLP_STAMP(1, sn++);
    if (tile + (int)gridDim.x < tiles) {
      request_bwd(T, g, (tile + gridDim.x) * ROWS, M, tid, N3p);
      request_x(T, g, (tile + gridDim.x) * ROWS, M, tid, K0p);
    }
    tile_backward(lds, g, W, G, pg, want_dx, tile * ROWS, M, tid, nm1, nb3);
    MLP_STAMP(1, sn++);
  }
  if (g.tq1 != nullptr) {
    // ---- this workgroup's share of the loss: the rows' squares summed in row order (the tile's LDS is free), then
    // one entry of the slab row (summed with the others in workgroup order) or, without a slab, one atomic addition
    const int dr = tid / N3p;
    if (dr < ROWS && tid == dr * N3p) lds[dr] = sq;
    lds_barrier();
    if (tid == 0) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) s += lds[r];
      s *= g.inv_n;
      if (pg && g.slab != nullptr)
        g.slab[(long)blockIdx.x * slab_floats(K0, N3) + slab_loss(K0, N3)] = s;
      else
        unsafeAtomicAdd(g.tloss, s);
    }
  }
  if (!pg) return;
  // ---- this workgroup's share of the parameter gradients, handed over once
  if (g.slab != nullptr) {
    float* sl = g.slab + (long)blockIdx.x * slab_floats(K0, N3);
#pragma unroll
    for (int c = 0; c < NMAX / 16; ++c)
      if (c < nb3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = 16 * c + 4 * q + e;
          if (n < N3) sl[slab_w3(K0, N3) + n * H + col0 + j] = G.w3[c][e];
        }
      }
#pragma unroll
    for (int c = 0; c < H / 16; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) sl[slab_w2(K0, N3) + (col0 + 4 * q + e) * H + 16 * c + j] = G.w2[c][e];
#pragma unroll
    for (int c = 0; c < NBX; ++c)
      if (c < nm1 && 16 * c + j < K0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) sl[slab_w1(K0, N3) + (col0 + 4 * q + e) * K0 + 16 * c + j] = G.w1[c][e];
      }
    if (tid < H) {
      sl[slab_b1(K0, N3) + tid] = G.b1;
      sl[slab_b2(K0, N3) + tid] = G.b2;
    }
    if (tid < N3) sl[slab_b3(K0, N3) + tid] = G.b3;
    return;
  }
#pragma unroll
  for (int c = 0; c < NMAX / 16; ++c)
    if (c < nb3) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = 16 * c + 4 * q + e;
        if (n < N3) unsafeAtomicAdd(&g.gw3[n * H + col0 + j], G.w3[c][e]);
      }
    }
#pragma unroll
  for (int c = 0; c < H / 16; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) unsafeAtomicAdd(&g.gw2[(col0 + 4 * q + e) * H + 16 * c + j], G.w2[c][e]);
#pragma unroll
  for (int c = 0; c < NBX; ++c)
    if (c < nm1 && 16 * c + j < K0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) unsafeAtomicAdd(&g.gw1[(long)(col0 + 4 * q + e) * K0 + 16 * c + j], G.w1[c][e]);
    }
  if (tid < H) {
    unsafeAtomicAdd(&g.gb1[tid], G.b1);
    unsafeAtomicAdd(&g.gb2[tid], G.b2);
  }
  if (tid < N3) unsafeAtomicAdd(&g.gb3[tid], G.b3);
}

// Optimizer step that rides on the slab sum (p == nullptr: none): the flat parameter, moment and target buffers
// share the layout of the flat gradient buffer at gbase, which holds every gradient tensor of the group.
struct AdamRide {
  const float* gbase;
  float *p, *m, *v, *target;
  const float* hyper;       // the eight scalars of the step, in device memory
  float tau, om;
};

// Sum of the slab rows in workgroup order, added to the gradient buffers.  A workgroup owns 64 consecutive
// parameters; its sixteen groups of 16 threads each sum every sixteenth row, 16 bytes per thread and row (256
// contiguous bytes per group and row, eight loads in flight per thread), the sixteen sums meet in LDS and are added in
// group order by the workgroup's first 64 threads, one parameter each.  blockIdx.y = network.
// ADAM: the thread that completes a gradient element also applies the Adam step to its parameter (and moves the
// target network's copy): the element's parameter and moments are requested before the slab rows, so they are
// there when the sum is, and the optimizer takes no launch and no pass of its own.
template <bool ADAM>
__global__ __launch_bounds__(256) void mlp3_slab_reduce_kernel(NetGroup grp, int rows, AdamRide ar) {
  __shared__ float part[16][65];
  const Net& g = grp.in[blockIdx.y];
  if (g.slab == nullptr || g.gw1 == nullptr) return;
  const int K0 = g.K0, N3 = g.N3;
  const int P = slab_floats(K0, N3), n_real = slab_loss(K0, N3) + (g.tq1 != nullptr ? 1 : 0);
  const int c = threadIdx.x & 15, grp16 = threadIdx.x >> 4, e4 = blockIdx.x * 64 + 4 * c;
  const int e = blockIdx.x * 64 + threadIdx.x;          // the element this thread completes (first 64 threads)
  const bool is_loss = e == slab_loss(K0, N3);          // the row's last entry: a share of the loss, when one is formed
  const bool last = threadIdx.x < 64 && e < n_real;
  float* dst = nullptr;
  float g0 = 0.f, p0 = 0.f, m0 = 0.f, v0 = 0.f, t0 = 0.f;
  long off = 0;
  if (last) {
    dst = e < slab_b1(K0, N3)   ? g.gw1 + e
          : e < slab_w2(K0, N3) ? g.gb1 + (e - slab_b1(K0, N3))
          : e < slab_b2(K0, N3) ? g.gw2 + (e - slab_w2(K0, N3))
          : e < slab_w3(K0, N3) ? g.gb2 + (e - slab_b2(K0, N3))
          : e < slab_b3(K0, N3) ? g.gw3 + (e - slab_w3(K0, N3))
          : !is_loss            ? g.gb3 + (e - slab_b3(K0, N3))
                                : g.tloss;
    g0 = *dst;
    if (ADAM && !is_loss) {
      off = dst - ar.gbase;
      p0 = ar.p[off];
      m0 = ar.m[off];
      v0 = ar.v[off];
      if (ar.target != nullptr) t0 = ar.target[off];
    }
  }
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e4 < P) {                                          // (P is a multiple of four: the 16 bytes lie inside the row)
    const float* p = g.slab + e4;
    int r = grp16;
    for (; r + 112 < rows; r += 128) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (long)(r + 16 * u) * P);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; r < rows; r += 16) {
      const float4 v = *reinterpret_cast<const float4*>(p + (long)r * P);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  part[grp16][4 * c] = s.x;
  part[grp16][4 * c + 1] = s.y;
  part[grp16][4 * c + 2] = s.z;
  part[grp16][4 * c + 3] = s.w;
  __syncthreads();
  if (last) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][threadIdx.x];
    const float gn = g0 + t;
    *dst = gn;
    if (ADAM && !is_loss) {
      AdamK k;
      k.step_size = ar.hyper[0]; k.beta1 = ar.hyper[1]; k.beta2 = ar.hyper[2]; k.one_minus_b1 = ar.hyper[3];
      k.one_minus_b2 = ar.hyper[4]; k.bc2_sqrt = ar.hyper[5]; k.eps = ar.hyper[6]; k.gscale = ar.hyper[7];
      adam_one(p0, gn, m0, v0, k);
      ar.p[off] = p0;
      ar.m[off] = m0;
      ar.v[off] = v0;
      if (ar.target != nullptr) ar.target[off] = polyak_one(t0, p0, ar.tau, ar.om);
    }
  }
}

// Workgroups per launch, shared among the networks of the group; a workgroup loops over its tiles.  The backward
// kernel and the forward kernel that also forms the input gradient hold one workgroup per CU (registers): as many
// as fill the chip once, 256.  The plain forward kernel fits twice (110 registers, 57 KB of LDS), and two workgroups
// per CU hide each other's prologue and barriers: 512.  The tiles are dealt evenly: with `per` workgroups allowed
// per network a workgroup takes ceil(tiles / per) tiles, and only as many workgroups are started as that takes
// (256 tiles on 85 workgroups would leave one of them a fourth tile while the rest have three).
// HHRL_MLP3_WGS / HHRL_MLP3_FWD_WGS set other totals (for sweeps).
static int grid_x(int tiles, int n, bool plain_forward = false) {
  static int total = 0, total_fwd = 0;
  if (total == 0) {
    const char* e = getenv("HHRL_MLP3_WGS");
    total = (e && atoi(e) > 0) ? atoi(e) : 256;
    const char* f = getenv("HHRL_MLP3_FWD_WGS");
    total_fwd = (f && atoi(f) > 0) ? atoi(f) : 2 * total;
  }
  int per = (plain_forward ? total_fwd : total) / n;
  if (per < 1) per = 1;
  const int rounds = ceil_div(tiles, per);
  return ceil_div(tiles, rounds);
}

// Fewest workgroups per network from which the slab hand-over is used (when the caller gave a slab): always.  With few
// rows the second launch costs what the stand-alone optimizer launch would (the step rides on it), and the gradients are
// the same bits on every run at every batch size.
constexpr int kSlabMinRows = 1;

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int fill_group(NetGroup& grp, const hhrl_mlp3_desc* d, int n, int M, bool backward) {
  HHRL_REQUIRE(d && n >= 1 && n <= kMaxNets, "mlp3: group of %d networks (1..%d supported)", n, kMaxNets);
  HHRL_REQUIRE(M > 0, "mlp3: M=%d", M);
  for (int i = 0; i < n; ++i) {
    const hhrl_mlp3_desc& s = d[i];
    HHRL_REQUIRE(s.x && s.w1 && s.b1 && s.w2 && s.b2 && s.w3 && s.b3, "mlp3: network %d: null pointer", i);
    HHRL_REQUIRE(hhrl_mlp3_supported(s.K0, H, H, s.N3), "mlp3: network %d: K0=%d N3=%d outside the fused kernels' range", i, s.K0, s.N3);
    HHRL_REQUIRE(s.ldx >= s.K0, "mlp3: network %d: ldx=%d < K0=%d", i, s.ldx, s.K0);
    HHRL_REQUIRE(aligned16(s.w2) && aligned16(s.w3), "mlp3: network %d: weights must be 16-byte aligned", i);
    HHRL_REQUIRE(s.act3 >= HHRL_ACT_NONE && s.act3 <= HHRL_ACT_SIGMOID, "mlp3: network %d: bad activation %d", i, s.act3);
    HHRL_REQUIRE((s.h1 == nullptr) == (s.h2 == nullptr), "mlp3: network %d: h1 and h2 go together", i);
    HHRL_REQUIRE(!s.h1 || (aligned16(s.h1) && aligned16(s.h2)), "mlp3: network %d: h1/h2 must be 16-byte aligned", i);
    if (!backward) {
      HHRL_REQUIRE(s.y && s.ldy >= s.N3, "mlp3_forward: network %d: y missing or ldy=%d < N3", i, s.ldy);
      HHRL_REQUIRE(!s.smooth_eps || (s.y2 && s.ldy2 >= s.N3), "mlp3_forward: network %d: smoothing needs y2 with ldy2 >= N3", i);
      HHRL_REQUIRE(!s.forward_dx || (s.dx && s.lddx >= s.K0), "mlp3_forward: network %d: forward_dx needs dx with lddx >= K0", i);
    } else {
      HHRL_REQUIRE(s.h1, "mlp3_backward: network %d: needs the hidden activations saved by the forward call", i);
      HHRL_REQUIRE(!s.dy || s.lddy >= s.N3, "mlp3_backward: network %d: lddy=%d < N3", i, s.lddy);
      HHRL_REQUIRE(s.act3 == HHRL_ACT_NONE || (s.y && s.ldy >= s.N3), "mlp3_backward: network %d: the activation gradient needs y", i);
      const bool any = s.gw1 || s.gb1 || s.gw2 || s.gb2 || s.gw3 || s.gb3;
      const bool all = s.gw1 && s.gb1 && s.gw2 && s.gb2 && s.gw3 && s.gb3;
      HHRL_REQUIRE(any == all, "mlp3_backward: network %d: give all six gradient pointers or none", i);
      HHRL_REQUIRE(all || s.dx, "mlp3_backward: network %d: nothing to compute", i);
      HHRL_REQUIRE(!s.dx || s.lddx >= s.K0, "mlp3_backward: network %d: lddx=%d < K0", i, s.lddx);
    }
    Net& g = grp.in[i];
    g.x = s.x; g.w1 = s.w1; g.b1 = s.b1; g.w2 = s.w2; g.b2 = s.b2; g.w3 = s.w3; g.b3 = s.b3;
    g.h1 = s.h1; g.h2 = s.h2; g.y = s.y; g.dy = s.dy;
    g.gw1 = s.gw1; g.gb1 = s.gb1; g.gw2 = s.gw2; g.gb2 = s.gb2; g.gw3 = s.gw3; g.gb3 = s.gb3; g.dx = s.dx;
    g.dy_const = s.dy_const;
    g.smooth_eps = backward ? nullptr : s.smooth_eps; g.y2 = s.y2; g.ldy2 = s.ldy2;
    g.ldx = s.ldx; g.ldy = s.ldy; g.lddy = s.lddy; g.lddx = s.lddx; g.K0 = s.K0; g.N3 = s.N3; g.act3 = s.act3;
    g.forward_dx = backward ? 0 : s.forward_dx;
    g.slab = nullptr;
    const bool tgt = backward && s.mse_q1t != nullptr;
    if (tgt) {
      HHRL_REQUIRE(s.N3 == 1 && s.y && !s.dy, "mlp3_backward: network %d: the TD3 target form needs N3 = 1, y and no dy", i);
      HHRL_REQUIRE(s.mse_q2t && s.mse_r && s.mse_loss && s.mse_ldr >= 1, "mlp3_backward: network %d: incomplete TD3 target description", i);
    }
    g.tq1 = tgt ? s.mse_q1t : nullptr; g.tq2 = s.mse_q2t; g.tr = s.mse_r; g.tloss = s.mse_loss; g.ty = s.mse_target;
    g.ldr = s.mse_ldr; g.gamma = s.mse_gamma; g.inv_n = s.mse_inv_n;
  }
  return HHRL_OK;
}

}  // namespace mf
}  // namespace hhrl

using namespace hhrl;
using namespace hhrl::mf;

extern "C" {

int hhrl_mlp3_supported(int K0, int H1, int H2, int N3) {
  return K0 >= 1 && K0 <= KMAX && H1 == H && H2 == H && N3 >= 1 && N3 <= NMAX;
}

long hhrl_mlp3_slab_floats(int K0, int N3, int M) {
  if (!hhrl_mlp3_supported(K0, H, H, N3) || M <= 0) return 0;
  const int tiles = ceil_div(M, ROWS), gx = grid_x(tiles, 1);     // a group of one has the most workgroups per network
  return gx >= kSlabMinRows ? (long)gx * slab_floats(K0, N3) : 0;
}

int hhrl_mlp3_forward_group(const hhrl_mlp3_desc* d, int n, int M, void* stream) {
  NetGroup grp;
  HHRL_TRY(fill_group(grp, d, n, M, false));
  hipStream_t st = as_stream(stream);
  double flops = 0.0;
  bool dx = false;
  for (int i = 0; i < n; ++i) {
    flops += 2.0 * M * ((double)d[i].K0 * H + (double)H * H + (double)H * d[i].N3);
    dx = dx || d[i].forward_dx != 0;
  }
  const int tiles = ceil_div(M, ROWS);
  const int gxf = grid_x(tiles, n, !dx);
  ProfScope ps("mlp3_forward_kernel", st, flops, 0.0);
  if (dx)
    hipLaunchKernelGGL(mlp3_forward_kernel<true>, dim3(gxf, n), dim3(NT), 0, st, grp, M, tiles);
  else
    hipLaunchKernelGGL(mlp3_forward_kernel<false>, dim3(gxf, n), dim3(NT), 0, st, grp, M, tiles);
  return launch_status("mlp3_forward_kernel");
}

// ride == nullptr: plain backward pass.  Otherwise the optimizer step of `ride` follows the gradients: inside the slab
// sum when every network of the group hands its gradients over as slab rows, else as a launch of its own.
static int mlp3_backward_impl(const hhrl_mlp3_desc* d, int n, int M, const hhrl_adam_ride* ride, void* stream) {
  NetGroup grp;
  HHRL_TRY(fill_group(grp, d, n, M, true));
  hipStream_t st = as_stream(stream);
  double flops = 0.0;
  for (int i = 0; i < n; ++i) flops += 4.0 * M * ((double)d[i].K0 * H + (double)H * H + (double)H * d[i].N3);
  const int tiles = ceil_div(M, ROWS), gx = grid_x(tiles, n);
  int n_slabs = 0;
  int pmax = 0;
  for (int i = 0; i < n; ++i) {
    const int P = slab_floats(d[i].K0, d[i].N3);
    if (d[i].gslab != nullptr && d[i].gw1 != nullptr && gx >= kSlabMinRows) {
      HHRL_REQUIRE(d[i].gslab_floats >= (long)gx * P, "mlp3_backward: network %d: slab of %ld floats, %ld needed (hhrl_mlp3_slab_floats)",
                   i, d[i].gslab_floats, (long)gx * P);
      HHRL_REQUIRE(aligned16(d[i].gslab), "mlp3_backward: network %d: the slab must be 16-byte aligned", i);
      grp.in[i].slab = d[i].gslab;
      ++n_slabs;
      pmax = P > pmax ? P : pmax;
    }
  }
  AdamRide ar{};
  if (ride != nullptr) {
    HHRL_REQUIRE(ride->grads && ride->params && ride->m && ride->v && ride->hyper_dev && ride->n > 0,
                 "mlp3_backward_adam: null pointer or n=%ld", ride->n);
    for (int i = 0; i < n; ++i) {
      const hhrl_mlp3_desc& s = d[i];
      HHRL_REQUIRE(s.gw1 != nullptr, "mlp3_backward_adam: network %d has no parameter gradients", i);
      const float* gp[6] = {s.gw1, s.gb1, s.gw2, s.gb2, s.gw3, s.gb3};
      const long gl[6] = {(long)H * s.K0, H, (long)H * H, H, (long)s.N3 * H, s.N3};
      for (int t = 0; t < 6; ++t)
        HHRL_REQUIRE(gp[t] >= ride->grads && gp[t] + gl[t] <= ride->grads + ride->n,
                     "mlp3_backward_adam: network %d: gradient tensor %d lies outside the flat gradient buffer", i, t);
    }
    ar.gbase = ride->grads; ar.p = ride->params; ar.m = ride->m; ar.v = ride->v; ar.target = ride->target;
    ar.hyper = ride->hyper_dev; ar.tau = (float)ride->tau; ar.om = (float)(1.0 - ride->tau);
  }
  {
    ProfScope ps("mlp3_backward_kernel", st, flops, 0.0);
    hipLaunchKernelGGL(mlp3_backward_kernel, dim3(gx, n), dim3(NT), 0, st, grp, M, tiles);
    HHRL_TRY(launch_status("mlp3_backward_kernel"));
  }
  const bool fused = ride != nullptr && n_slabs == n;
  if (n_slabs > 0) {
    ProfScope ps("mlp3_slab_reduce_kernel", st, 0.0, 4.0 * n * (double)pmax * (gx + 2 + (fused ? 7 : 0)));
    if (fused)
      hipLaunchKernelGGL(mlp3_slab_reduce_kernel<true>, dim3(ceil_div(pmax, 64), n), dim3(256), 0, st, grp, gx, ar);
    else
      hipLaunchKernelGGL(mlp3_slab_reduce_kernel<false>, dim3(ceil_div(pmax, 64), n), dim3(256), 0, st, grp, gx, ar);
    HHRL_TRY(launch_status("mlp3_slab_reduce_kernel"));
  }
  if (ride != nullptr && !fused) {
    if (ride->target != nullptr)
      return hhrl_adam_polyak_step_dev(ride->params, ride->grads, ride->m, ride->v, ride->n, ride->hyper_dev, ride->target,
                                       ride->tau, stream);
    return hhrl_adam_step_dev(ride->params, ride->grads, ride->m, ride->v, ride->n, ride->hyper_dev, stream);
  }
  return HHRL_OK;
}

int hhrl_mlp3_backward_group(const hhrl_mlp3_desc* d, int n, int M, void* stream) {
  return mlp3_backward_impl(d, n, M, nullptr, stream);
}

int hhrl_mlp3_backward_group_adam(const hhrl_mlp3_desc* d, int n, int M, const hhrl_adam_ride* ride, void* stream) {
  HHRL_REQUIRE(ride != nullptr, "mlp3_backward_group_adam: null optimizer description");
  return mlp3_backward_impl(d, n, M, ride, stream);
}

#ifdef HHRL_STAMPS
// diagnostic builds only: copies the recorded clocks (forward, backward: 64 stamps each) to the host
int hhrl_debug_mlp_stamps(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(hhrl::mf::g_mlp_stamps), sizeof(long long) * 2 * 64) == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
