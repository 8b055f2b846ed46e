// Three dense layers (Linear+ReLU, Linear+ReLU, Linear+act) as ONE kernel per direction, for the
// 128-wide networks of the worker agents (WorkerActor / WorkerCritic, algorithm.py:1187-1258, hidden
// sizes from configs.py:159): a batch-64 update is bound by the number of launches, and these
// networks fit a workgroup's LDS whole.
//
// A workgroup owns 32 rows of one network (blockIdx.y = network of the group: twin Q heads, target
// networks ... run side by side).  All three weight matrices are staged into LDS images once, the
// activations of the chain stay in LDS; the backward kernel re-reads the saved hidden activations,
// overwrites them with their gradients in place and adds the weight gradients of its rows to the
// gradient buffers with float atomics.
#include "common.h"
#include "mfma_tiles.h"

namespace hhrl {
namespace mf {

using namespace mt;

constexpr int H = 128;          // both hidden widths
constexpr int KMAX = 64;        // widest input
constexpr int NMAX = 32;        // widest output
constexpr int kMaxNets = HHRL_MLP3_GROUP_MAX;

struct Net {
  const float* x;
  const float *w1, *b1, *w2, *b2, *w3, *b3;
  float *h1, *h2, *y;
  const float* dy;
  float *gw1, *gb1, *gw2, *gb2, *gw3, *gb3, *dx;
  const float* smooth_eps;
  float* y2;
  float dy_const;
  int ldx, ldy, lddy, lddx, K0, N3, act3, ldy2, forward_dx;
};
struct NetGroup { Net in[kMaxNets]; };

// LDS map (floats)
constexpr int L_W1 = 0;                         // [128][65]
constexpr int L_W2 = L_W1 + H * (KMAX + 1);     // [128][129]
constexpr int L_W3 = L_W2 + H * (H + 1);        // [32][129]   rows >= N3 zero
constexpr int L_B = L_W3 + NMAX * (H + 1);      // b1[128] b2[128] b3[32]
constexpr int L_X = L_B + 2 * H + NMAX;         // [32][65]    columns >= K0 zero
constexpr int L_A1 = L_X + 32 * (KMAX + 1);     // [32][129]
constexpr int L_A2 = L_A1 + 32 * (H + 1);       // [32][129]
constexpr int L_Y = L_A2 + 32 * (H + 1);        // [32][33]
constexpr int L_END = L_Y + 32 * 33;
static_assert(L_END * 4 <= 163840, "the three-layer kernels must fit the 160 KiB LDS");

// acc[t] += A[32 x K] * B[K x 32] with a run-time reduction length (a multiple of 2)
template <int TJ, int ASI, int ASK, int BSK, int BSJ>
__device__ __forceinline__ void wave_mma_rt(f32x16 (&acc)[TJ], const float* a, const float* b, int lane, int K) {
  const int r = lane & 31, h = lane >> 5;
  const float* ap = a + r * ASI + h * ASK;
  const float* bp = b + h * BSK + r * BSJ;
  for (int k = 0; k < K; k += 2) {
    const float av = ap[k * ASK];
#pragma unroll
    for (int t = 0; t < TJ; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[k * BSK + t * 32 * BSJ], acc[t], 0, 0, 0);
  }
}

// ---- staging helpers: every global load of a stage is issued before its first LDS write.
//      NT = threads of the workgroup: 256, or 512 when four more waves share the copies (at one wave per
//      SIMD the copies are bound by instruction issue); the matrix phases stay on waves 0..3.
template <int NT>
struct StageW {
  float4 w2[W4Map<H, H, NT>::STEPS];        // [128][128]
  float4 w3[W4Map<NMAX, H, NT>::STEPS];     // [N3][128], rows >= N3 zero
};
template <int NT>
__device__ __forceinline__ void request_w23(StageW<NT>& R, const Net& g, int tid) {
  request_w4<H, H, W4Map<H, H, NT>::STEPS, NT>(R.w2, g.w2, tid);
#pragma unroll
  for (int u = 0; u < W4Map<NMAX, H, NT>::STEPS; ++u) {
    int row, col;
    W4Map<NMAX, H, NT>::pos(tid, u, row, col);
    R.w3[u] = row < g.N3 ? *reinterpret_cast<const float4*>(g.w3 + row * H + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int NT>
__device__ __forceinline__ void commit_w23(float* lds, const StageW<NT>& R, int tid) {
  commit_w4<H, H, W4Map<H, H, NT>::STEPS, NT>(lds + L_W2, H + 1, R.w2, tid);
  commit_w4<NMAX, H, W4Map<NMAX, H, NT>::STEPS, NT>(lds + L_W3, H + 1, R.w3, tid);
}
// W1 [128][K0] (any K0 <= 64) and the x tile [32][K0]: element-wise, zero padding up to the even width K0p
template <int NT>
__device__ __forceinline__ void stage_w1_x(float* lds, const Net& g, int row0, int M, int tid, bool want_w1) {
  const int K0 = g.K0, K0p = (K0 + 7) & ~7;
  if (want_w1) {
    for (int base = 0; base < H * K0p; base += 8 * NT) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = base + u * NT + tid, rr = e / K0p, c = e - rr * K0p;
        v[u] = (e < H * K0p && c < K0) ? g.w1[rr * K0 + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = base + u * NT + tid, rr = e / K0p, c = e - rr * K0p;
        if (e < H * K0p) lds[L_W1 + rr * (KMAX + 1) + c] = v[u];
      }
    }
  }
  float v[2048 / NT];
#pragma unroll
  for (int u = 0; u < 2048 / NT; ++u) {
    const int e = u * NT + tid, rr = e / K0p, c = e - rr * K0p;
    v[u] = (e < 32 * K0p && c < K0 && row0 + rr < M) ? g.x[(long)(row0 + rr) * g.ldx + c] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < 2048 / NT; ++u) {
    const int e = u * NT + tid, rr = e / K0p, c = e - rr * K0p;
    if (e < 32 * K0p) lds[L_X + rr * (KMAX + 1) + c] = v[u];
  }
}
// a [32][128] tile of a row-major [M][128] matrix <-> an image with row stride 129
template <int NT, int STEPS>
__device__ __forceinline__ void request_rows128(float4 (&v)[STEPS], const float* src, int row0, int M, int tid) {
  static_assert(STEPS == W4Map<32, H, NT>::STEPS, "register array does not match the tile");
#pragma unroll
  for (int u = 0; u < STEPS; ++u) {
    int row, col;
    W4Map<32, H, NT>::pos(tid, u, row, col);
    v[u] = row0 + row < M ? *reinterpret_cast<const float4*>(src + (long)(row0 + row) * H + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int NT>
__device__ __forceinline__ void store_rows128(const float* img, float* dst, int row0, int M, int tid) {
#pragma unroll
  for (int u = 0; u < 4096 / NT; ++u) {
    const int e = u * NT + tid, rr = e >> 7, c = e & 127;
    if (row0 + rr < M) dst[(long)(row0 + rr) * H + c] = img[rr * (H + 1) + c];
  }
}

// ------------------------------------------------------------------ forward
template <int NT>
__global__ __launch_bounds__(NT, 1) void mlp3_forward_kernel(NetGroup grp, int M) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const Net& g = grp.in[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const bool mat = wave < 4;                      // this wave takes part in the matrix phases
  const int row0 = blockIdx.x * 32;
  const int K0p = (g.K0 + 7) & ~7;
  float* bs = lds + L_B;
  {
    StageW<NT> R;
    request_w23(R, g, tid);
    const float b1 = tid < H ? g.b1[tid] : (tid < 2 * H ? g.b2[tid - H] : 0.f);
    const float b3 = tid < g.N3 ? g.b3[tid] : 0.f;
    stage_w1_x<NT>(lds, g, row0, M, tid, true);
    commit_w23(lds, R, tid);
    if (tid < 2 * H) bs[tid] = b1;
    if (tid < NMAX) bs[2 * H + tid] = b3;
  }
  __syncthreads();
  // ---- layer 1 (K0 -> 128, ReLU): one 32-column tile per wave
  if (mat) {
    f32x16 acc[1];
    acc[0] = splat(bs[32 * wave + r]);
    wave_mma_rt<1, KMAX + 1, 1, 1, KMAX + 1>(acc, lds + L_X, lds + L_W1 + 32 * wave * (KMAX + 1), lane, K0p);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] = fmaxf(acc[0][i], 0.f);
    store_tile<H + 1>(lds + L_A1, 32 * wave, acc[0], lane);
  }
  __syncthreads();
  // ---- layer 2 (128 -> 128, ReLU)
  if (mat) {
    f32x16 acc[1];
    acc[0] = splat(bs[H + 32 * wave + r]);
    wave_mma<1, H, H + 1, 1, 1, H + 1>(acc, lds + L_A1, lds + L_W2 + 32 * wave * (H + 1), lane);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] = fmaxf(acc[0][i], 0.f);
    store_tile<H + 1>(lds + L_A2, 32 * wave, acc[0], lane);
  }
  __syncthreads();
  // ---- layer 3 (128 -> N3 <= 32, act3): one 16x16 block per wave, straight to global memory
  if (mat) {
    const int rb = wave & 1, cb = wave >> 1, j = lane & 15, q = lane >> 4;
    const f32x4 c = wave_mma16<H, H + 1, 1, 1, H + 1>(lds + L_A2 + 16 * rb * (H + 1), lds + L_W3 + 16 * cb * (H + 1), lane,
                                                     bs[2 * H + 16 * cb + j]);
    const int col = 16 * cb + j;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = row0 + 16 * rb + 4 * q + e;
      if (row < M && col < g.N3) {
        const float v = act_apply(c[e], g.act3);
        g.y[(long)row * g.ldy + col] = v;
        if (g.smooth_eps != nullptr) {      // target-policy smoothing of the same value, as td3_smooth_action_kernel
          float n = 0.2f * g.smooth_eps[(long)row * g.N3 + col];
          n = fminf(fmaxf(n, -0.5f), 0.5f);
          g.y2[(long)row * g.ldy2 + col] = fminf(fmaxf(v + n, -1.f), 1.f);
        }
      }
    }
  }
  if (g.forward_dx) {
    // ---- the input gradient for dL/dy = dy_const on every row, in the same launch (the policy step's d(-mean Q)/d(action):
    //      the network's weights and activations are in LDS already).  Same products as the backward kernel without the
    //      parameter gradients; the hidden activations are overwritten by their gradients and not written to memory.
    float* D3 = lds + L_Y;
    __syncthreads();                                   // layer 3 has read h2
#pragma unroll
    for (int u = 0; u < 1024 / NT; ++u) {
      const int e = u * NT + tid, rr = e >> 5, c = e & 31;
      float v = 0.f;
      if (c < g.N3 && row0 + rr < M) {
        v = g.dy_const;
        if (g.act3 != HHRL_ACT_NONE) v *= act_grad_from_output(g.y[(long)(row0 + rr) * g.ldy + c], g.act3);
      }
      D3[rr * 33 + c] = v;
    }
    __syncthreads();
    f32x16 d[1];
    if (mat) {                                         // dh2 = (D3 W3) o relu'(h2), column block `wave`, in place
      d[0] = splat(0.f);
      wave_mma<1, 32, 33, 1, H + 1, 1>(d, D3, lds + L_W3 + 32 * wave, lane);
      wave_sync();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float* p = lds + L_A2 + drow(i, h) * (H + 1) + 32 * wave + r;
        *p = *p > 0.f ? d[0][i] : 0.f;
      }
    }
    __syncthreads();
    if (mat) {                                         // dh1 = (dh2 W2) o relu'(h1)
      d[0] = splat(0.f);
      wave_mma<1, H, H + 1, 1, H + 1, 1>(d, lds + L_A2, lds + L_W2 + 32 * wave, lane);
    }
    __syncthreads();                                   // every wave is done reading dh2 ... (and nobody reads h1 any more)
    if (mat) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float* p = lds + L_A1 + drow(i, h) * (H + 1) + 32 * wave + r;
        *p = *p > 0.f ? d[0][i] : 0.f;
      }
    }
    __syncthreads();
    {                                                  // dx = dh1 W1: 16x16 blocks over the helper waves (or waves 0..3)
      const int xw = NT == 512 ? wave - 4 : wave;
      if (xw >= 0) {
        const int rb = xw & 1, j = lane & 15, q = lane >> 4;
        for (int cb = xw >> 1; 16 * cb < K0p; cb += 2) {
          const f32x4 c = wave_mma16<H, H + 1, 1, KMAX + 1, 1>(lds + L_A1 + 16 * rb * (H + 1), lds + L_W1 + 16 * cb, lane, 0.f);
          const int col = 16 * cb + j;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int row = row0 + 16 * rb + 4 * q + e;
            if (row < M && col < g.K0) g.dx[(long)row * g.lddx + col] = c[e];
          }
        }
      }
    }
    return;
  }
  if (g.h1 != nullptr) {
    store_rows128<NT>(lds + L_A1, g.h1, row0, M, tid);
    store_rows128<NT>(lds + L_A2, g.h2, row0, M, tid);
  }
  (void)h;
}

// ------------------------------------------------------------------ backward
// With helper waves (NT == 512) waves 4..7 share the copies and own the bias gradients (thread T_B + column).
template <int NT>
__global__ __launch_bounds__(NT, 1) void mlp3_backward_kernel(NetGroup grp, int M) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const Net& g = grp.in[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const bool mat = wave < 4;                      // this wave takes part in the matrix phases
  constexpr int T_B = NT == 512 ? 256 : 0;
  const int bcol = tid - T_B;                     // bias-gradient column of this thread (if in range)
  const int row0 = blockIdx.x * 32;
  const int K0 = g.K0, K0p = (K0 + 7) & ~7, N3 = g.N3;
  const bool pg = g.gw1 != nullptr;            // parameter gradients wanted
  const bool want_dx = g.dx != nullptr;
  float* W1s = lds + L_W1;
  float* W2s = lds + L_W2;
  float* W3s = lds + L_W3;
  float* Xs = lds + L_X;
  float* A1 = lds + L_A1;     // h1, then its gradient
  float* A2 = lds + L_A2;     // h2, then its gradient
  float* D3 = lds + L_Y;      // gradient at the pre-activation of layer 3, columns >= N3 zero
  {
    StageW<NT> R;
    float4 a1[W4Map<32, H, NT>::STEPS], a2[W4Map<32, H, NT>::STEPS];
    request_w23(R, g, tid);
    request_rows128<NT>(a1, g.h1, row0, M, tid);
    request_rows128<NT>(a2, g.h2, row0, M, tid);
    // gradient at the output through act3
    float d3[1024 / NT];
#pragma unroll
    for (int u = 0; u < 1024 / NT; ++u) {
      const int e = u * NT + tid, rr = e >> 5, c = e & 31;
      float v = 0.f;
      if (c < N3 && row0 + rr < M) {
        v = g.dy ? g.dy[(long)(row0 + rr) * g.lddy + c] : g.dy_const;
        if (g.act3 != HHRL_ACT_NONE) v *= act_grad_from_output(g.y[(long)(row0 + rr) * g.ldy + c], g.act3);
      }
      d3[u] = v;
    }
    stage_w1_x<NT>(lds, g, row0, M, tid, want_dx);
    commit_w23(lds, R, tid);
    commit_w4<32, H, W4Map<32, H, NT>::STEPS, NT>(A1, H + 1, a1, tid);
    commit_w4<32, H, W4Map<32, H, NT>::STEPS, NT>(A2, H + 1, a2, tid);
#pragma unroll
    for (int u = 0; u < 1024 / NT; ++u) {
      const int e = u * NT + tid;
      D3[(e >> 5) * 33 + (e & 31)] = d3[u];
    }
  }
  __syncthreads();
  // ---- layer 3: db3, dW3 [N3][128] += D3^T h2, dh2 = (D3 W3) o relu'(h2) in place (column block `wave`)
  if (pg && bcol >= 0 && bcol < N3) {
    float s = 0.f;
    for (int rr = 0; rr < 32; ++rr) s += D3[rr * 33 + bcol];
    unsafeAtomicAdd(&g.gb3[bcol], s);
  }
  if (mat) {
    if (pg) {
      f32x16 acc[1];
      acc[0] = splat(0.f);
      wave_mma<1, 32, 1, 33, H + 1, 1>(acc, D3, A2 + 32 * wave, lane);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = drow(i, h);
        if (row < N3) unsafeAtomicAdd(&g.gw3[row * H + 32 * wave + r], acc[0][i]);
      }
    }
    f32x16 d[1];
    d[0] = splat(0.f);
    wave_mma<1, 32, 33, 1, H + 1, 1>(d, D3, W3s + 32 * wave, lane);
    wave_sync();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float* p = A2 + drow(i, h) * (H + 1) + 32 * wave + r;
      *p = *p > 0.f ? d[0][i] : 0.f;
    }
  }
  __syncthreads();
  // ---- layer 2: db2, dW2 [128][128] += dh2^T h1 (row tile `wave`), dh1 = (dh2 W2) o relu'(h1) in place
  {
    if (pg && bcol >= 0 && bcol < H) {
      float s = 0.f;
      for (int rr = 0; rr < 32; ++rr) s += A2[rr * (H + 1) + bcol];
      unsafeAtomicAdd(&g.gb2[bcol], s);
    }
    f32x16 d[1];
    d[0] = splat(0.f);
    if (mat) {
      if (pg) {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = splat(0.f);
        wave_mma<4, 32, 1, H + 1, H + 1, 1>(acc, A2 + 32 * wave, A1, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) unsafeAtomicAdd(&g.gw2[(32 * wave + drow(i, h)) * H + 32 * t + r], acc[t][i]);
      }
      wave_mma<1, H, H + 1, 1, H + 1, 1>(d, A2, W2s + 32 * wave, lane);
    }
    __syncthreads();                                   // every wave is done reading h1
    if (mat) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float* p = A1 + drow(i, h) * (H + 1) + 32 * wave + r;
        *p = *p > 0.f ? d[0][i] : 0.f;
      }
    }
  }
  __syncthreads();
  // ---- layer 1: db1, dW1 [128][K0] += dh1^T x (row tile `wave`), dx = dh1 W1 (16x16 blocks)
  if (pg && bcol >= 0 && bcol < H) {
    float s = 0.f;
    for (int rr = 0; rr < 32; ++rr) s += A1[rr * (H + 1) + bcol];
    unsafeAtomicAdd(&g.gb1[bcol], s);
  }
  if (pg && mat) {
    f32x16 acc[2];
    acc[0] = splat(0.f);
    acc[1] = splat(0.f);
    wave_mma<2, 32, 1, H + 1, KMAX + 1, 1>(acc, A1 + 32 * wave, Xs, lane);     // columns >= K0p read zero padding or junk, dropped below
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int col = 32 * t + r;
      if (col < K0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) unsafeAtomicAdd(&g.gw1[(32 * wave + drow(i, h)) * K0 + col], acc[t][i]);
      }
    }
  }
  if (want_dx) {
    // 16x16 blocks of dx: over waves 0..3, or over the four helper waves when there are any (they are idle here)
    const int xw = NT == 512 ? wave - 4 : wave;
    if (xw >= 0) {
      const int rb = xw & 1, j = lane & 15, q = lane >> 4;
      for (int cb = xw >> 1; 16 * cb < K0p; cb += 2) {
        const f32x4 c = wave_mma16<H, H + 1, 1, KMAX + 1, 1>(A1 + 16 * rb * (H + 1), W1s + 16 * cb, lane, 0.f);
        const int col = 16 * cb + j;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = row0 + 16 * rb + 4 * q + e;
          if (row < M && col < K0) g.dx[(long)row * g.lddx + col] = c[e];
        }
      }
    }
  }
}

// Threads per workgroup: 512 (four helper waves) unless HHRL_MLP_THREADS=256 asks for the plain form.
static int g_threads = 0;
static int set_attrs() {
  const void* fns[] = {reinterpret_cast<const void*>(mlp3_forward_kernel<256>), reinterpret_cast<const void*>(mlp3_forward_kernel<512>),
                       reinterpret_cast<const void*>(mlp3_backward_kernel<256>), reinterpret_cast<const void*>(mlp3_backward_kernel<512>)};
  for (int i = 0; i < 4; ++i) HHRL_TRY(ensure_dyn_lds(fns[i], L_END * 4, "mlp3"));
  if (g_threads == 0) {
    const char* e = getenv("HHRL_MLP_THREADS");
    g_threads = (e && atoi(e) == 256) ? 256 : 512;
  }
  return HHRL_OK;
}

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

int hhrl_mlp3_forward_group(const hhrl_mlp3_desc* d, int n, int M, void* stream) {
  NetGroup grp;
  HHRL_TRY(fill_group(grp, d, n, M, false));
  HHRL_TRY(set_attrs());
  hipStream_t st = as_stream(stream);
  double flops = 0.0;
  for (int i = 0; i < n; ++i) flops += 2.0 * M * ((double)d[i].K0 * H + (double)H * H + (double)H * d[i].N3);
  ProfScope ps("mlp3_forward_kernel", st, flops, 0.0);
  if (g_threads == 512)
    hipLaunchKernelGGL(mlp3_forward_kernel<512>, dim3(ceil_div(M, 32), n), dim3(512), L_END * 4, st, grp, M);
  else
    hipLaunchKernelGGL(mlp3_forward_kernel<256>, dim3(ceil_div(M, 32), n), dim3(256), L_END * 4, st, grp, M);
  return launch_status("mlp3_forward_kernel");
}

int hhrl_mlp3_backward_group(const hhrl_mlp3_desc* d, int n, int M, void* stream) {
  NetGroup grp;
  HHRL_TRY(fill_group(grp, d, n, M, true));
  HHRL_TRY(set_attrs());
  hipStream_t st = as_stream(stream);
  double flops = 0.0;
  for (int i = 0; i < n; ++i) flops += 4.0 * M * ((double)d[i].K0 * H + (double)H * H + (double)H * d[i].N3);
  ProfScope ps("mlp3_backward_kernel", st, flops, 0.0);
  if (g_threads == 512)
    hipLaunchKernelGGL(mlp3_backward_kernel<512>, dim3(ceil_div(M, 32), n), dim3(512), L_END * 4, st, grp, M);
  else
    hipLaunchKernelGGL(mlp3_backward_kernel<256>, dim3(ceil_div(M, 32), n), dim3(256), L_END * 4, st, grp, M);
  return launch_status("mlp3_backward_kernel");
}

}  // extern "C"
