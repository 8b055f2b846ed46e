// Bidirectional single-layer LSTM over a short window, last time index only,
// followed by Linear(2H -> H): forward and back-propagation through time, for a
// GROUP of independent instances per call (the per-stream LSTMs of a critic,
// the actors and their targets): every stage is ONE launch for the whole group,
// because batch-64 updates are bound by the number of launches.  (Hidden sizes that
// are not a multiple of 32, or exceed 128, take a GEMM + cell-kernel pair per step.)
//
// Layout choices: the [B, T, o] observation window is transposed once to
// time-major [T, B, o], so that every time step is a contiguous [B, .] block.
// A forward time step is one kernel: input projection + recurrent product on
// the matrix cores, gate nonlinearities and the cell update in its epilogue.
// A backward time step is one kernel: dh_{t-1} = dgates_t * W_hh with the cell
// backward of step t-1 in its epilogue.  The weight gradients of the whole
// window are split-reduction GEMMs over the rows (t, b).
// Hidden and cell states are kept with a leading zero slot (slot t+1 = state
// after step t), so "state before step t" for all t is the same buffer
// without the last slot.
#include "common.h"

namespace hhrl {

static inline long align64(long n) { return (n + 63) & ~63L; }

struct LstmWs {
  float *x_tm, *gates, *hbuf, *cbuf, *gates_r, *c_r, *cat, *dgates, *dgates_r, *dcat, *dc, *dh;
  long total;
};

static LstmWs carve(float* ws, long B, long T, long o, long H) {
  LstmWs w;
  long off = 0;
  auto take = [&](long n) { float* p = ws ? ws + off : nullptr; off += align64(n); return p; };
  w.x_tm = take(T * B * o);
  w.gates = take(T * B * 4 * H);
  w.hbuf = take((T + 1) * B * H);
  w.cbuf = take((T + 1) * B * H);
  w.gates_r = take(B * 4 * H);
  w.c_r = take(B * H);
  w.cat = take(B * 2 * H);
  w.dgates = take(T * B * 4 * H);
  w.dgates_r = take(B * 4 * H);
  w.dcat = take(B * 2 * H);
  w.dc = take(B * H);
  w.dh = take(B * H);
  w.total = off;
  return w;
}

constexpr int kMaxInst = HHRL_LSTM_GROUP_MAX;

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int drow16(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ---- [B, T*o] rows with leading dimension ld -> [T, B, o]; also clears the zero slot of the state buffers
struct PrepInst { const float* in; float *out, *h0, *c0; int ld, o; };
struct PrepGroup { PrepInst in[kMaxInst]; };
__global__ void lstm_prep_kernel(PrepGroup grp, int B, int T, int H) {
  const PrepInst& g = grp.in[blockIdx.z];
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (long)B * H) {
    g.h0[idx] = 0.f;
    g.c0[idx] = 0.f;
  }
  if (idx >= (long)B * T * g.o) return;
  const int c = idx % g.o;
  const long bt = idx / g.o;
  const int t = bt % T, b = bt / T;
  g.out[((long)t * B + b) * g.o + c] = g.in[(long)b * g.ld + t * g.o + c];
}

// ---- one forward time step: gates = x_t W_ih^T + h_prev W_hh^T + b_ih + b_hh, then the cell.
// A workgroup owns 32 batch rows x 8 hidden units: a 32x32 tile of gate columns ordered
// (gate q, unit j) -> column 8*q + j.  The reduction runs over [h_prev | x_t] against
// [W_hh | W_ih] rows staged side by side in LDS; the four waves split it (short MFMA chains,
// many workgroups: the step is latency-bound), the partial tiles are summed through LDS in wave
// order, and each thread finishes one (row, unit) cell.  `gates` receives the ACTIVATED gates.
// h_prev == nullptr: the state before the step is zero (first step, and the single step of the
// reverse direction), only the input part is reduced.
struct StepFwdInst {
  const float *x, *w_ih, *w_hh, *b_ih, *b_hh, *h_prev, *c_prev;
  float *gates, *c_out, *h_out, *h_out2;
  int o, ldh, ldh2;
};
struct StepFwdGroup { StepFwdInst in[kMaxInst]; };
constexpr int kMaxObsPad = 64;      // widest per-step observation the fused step kernel stages

// Threads per workgroup of the forward step: 512 unless HHRL_LSTM_THREADS=256 asks for the plain form.
static int step_threads() {
  static int nt = 0;
  if (nt == 0) {
    const char* e = getenv("HHRL_LSTM_THREADS");
    nt = (e && atoi(e) == 256) ? 256 : 512;
  }
  return nt;
}
static long step_fwd_lds_bytes(int H) { return 4L * (2L * 32 * (H + kMaxObsPad + 1) + 4 * 32 * 33); }

// NT threads: 256, or 512 when four more waves share the copies into LDS (bound by instruction issue at
// one wave per SIMD); the product and the gate epilogue stay on waves 0..3.
template <int NT>
__global__ __launch_bounds__(NT) void lstm_step_forward_kernel(StepFwdGroup grp, int B, int H) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const StepFwdInst& g = grp.in[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int u0 = blockIdx.x * 8, row0 = blockIdx.y * 32;
  const int o = g.o, opad = (o + 7) & ~7;
  const int ld = H + kMaxObsPad + 1;
  float* Ah = lds;                       // [32][ld]  columns [0,H): h_prev rows, [H, H+opad): x_t rows
  float* Wq = lds + 32 * ld;             // [32][ld]  rows (gate q, unit j) at image row 8*q + j: W_hh | W_ih
  float* part = Wq + 32 * ld;            // [4][32][33] partial gate tiles
  const bool has_h = g.h_prev != nullptr;
  {
    // every global load is issued before the first LDS write
    const int hq = H >> 2;               // float4 per state / W_hh row (H % 32 == 0 on this path)
    float4 v[1024 / NT], wv[1024 / NT];
    if (has_h) {
#pragma unroll
      for (int it = 0; it < 1024 / NT; ++it) {    // 32*hq float4 each, <= 1024 / NT per thread for H <= 128
        const int i4 = it * NT + tid;
        const int rr = i4 / hq, c4 = i4 - rr * hq;
        const bool in = i4 < 32 * hq;
        v[it] = (in && row0 + rr < B) ? reinterpret_cast<const float4*>(g.h_prev + (long)(row0 + rr) * H)[c4]
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
        const int q = rr >> 3, j = rr & 7;
        wv[it] = in ? reinterpret_cast<const float4*>(g.w_hh + ((long)q * H + u0 + j) * H)[c4]
                    : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    // x_t tile (32 rows x o, contiguous in time-major storage) and the 32 W_ih rows: 32*opad elements each, <= 8 per thread
    float xv[2048 / NT], wi[2048 / NT];
#pragma unroll
    for (int it = 0; it < 2048 / NT; ++it) {
      const int e = it * NT + tid;
      const int rr = e / opad, c = e - rr * opad;
      const bool in = e < 32 * opad && c < o;
      xv[it] = (in && row0 + rr < B) ? g.x[(long)(row0 + rr) * o + c] : 0.f;
      const int q = rr >> 3, j = rr & 7;
      wi[it] = in ? g.w_ih[((long)q * H + u0 + j) * o + c] : 0.f;
    }
    if (has_h) {
#pragma unroll
      for (int it = 0; it < 1024 / NT; ++it) {
        const int i4 = it * NT + tid;
        const int rr = i4 / hq, c4 = i4 - rr * hq;
        if (i4 < 32 * hq) {
          float* p = Ah + rr * ld + 4 * c4;
          p[0] = v[it].x; p[1] = v[it].y; p[2] = v[it].z; p[3] = v[it].w;
          float* q = Wq + rr * ld + 4 * c4;
          q[0] = wv[it].x; q[1] = wv[it].y; q[2] = wv[it].z; q[3] = wv[it].w;
        }
      }
    }
#pragma unroll
    for (int it = 0; it < 2048 / NT; ++it) {
      const int e = it * NT + tid;
      const int rr = e / opad, c = e - rr * opad;
      if (e < 32 * opad) {
        Ah[rr * ld + H + c] = xv[it];
        Wq[rr * ld + H + c] = wi[it];
      }
    }
  }
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (wave < 4) {
    const int kb = has_h ? 0 : H, len = (has_h ? H : 0) + opad;   // a multiple of 8
    const int kq = len >> 2;                                      // this wave's quarter of the reduction
    const float* ap = Ah + r * ld + kb + wave * kq + h;
    const float* bp = Wq + r * ld + kb + wave * kq + h;
    for (int k = 0; k < kq; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k], bp[k], acc, 0, 0, 0);
  }
  if (wave < 4) {
#pragma unroll
    for (int i = 0; i < 16; ++i) part[(wave * 32 + drow16(i, h)) * 33 + r] = acc[i];
  }
  __syncthreads();
  if (wave < 4) {
    const int rr = tid >> 3, j = tid & 7;
    const int b = row0 + rr, u = u0 + j;
    if (b < B && u < H) {
      float pre[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = 8 * q + j;
        pre[q] = (part[(0 * 32 + rr) * 33 + col] + part[(1 * 32 + rr) * 33 + col]) +
                 (part[(2 * 32 + rr) * 33 + col] + part[(3 * 32 + rr) * 33 + col]);
        pre[q] += g.b_ih[q * H + u] + g.b_hh[q * H + u];
      }
      float* gr = g.gates + (long)b * 4 * H;
      const float gi = sigmoidf_(pre[0]);
      const float gf = sigmoidf_(pre[1]);
      const float gg = tanhf(pre[2]);
      const float go = sigmoidf_(pre[3]);
      gr[u] = gi; gr[H + u] = gf; gr[2 * H + u] = gg; gr[3 * H + u] = go;
      const float cp = g.c_prev ? g.c_prev[(long)b * H + u] : 0.f;
      const float c = gf * cp + gi * gg;
      const float hv = go * tanhf(c);
      g.c_out[(long)b * H + u] = c;
      g.h_out[(long)b * g.ldh + u] = hv;
      if (g.h_out2) g.h_out2[(long)b * g.ldh2 + u] = hv;
    }
  }
}

// ---- hidden sizes outside the fused step kernels (not a multiple of 32, or beyond 128): the step is a
//      GEMM accumulated onto the projected gates, then this cell kernel.  gates: [B,4H] pre-activations
//      (input projection + recurrent product), in place -> activated i,f,g,o.  bias2 (bias_hh) is added first.
__global__ void lstm_cell_forward_kernel(float* gates, const float* __restrict__ bias2,
                                         const float* __restrict__ c_prev, float* c_out, float* h_out, int ldh,
                                         float* h_out2, int ldh2, int B, int H) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, j = idx % H;
  float* gr = gates + (long)b * 4 * H;
  const float i = sigmoidf_(gr[j] + bias2[j]);
  const float f = sigmoidf_(gr[H + j] + bias2[H + j]);
  const float g = tanhf(gr[2 * H + j] + bias2[2 * H + j]);
  const float o = sigmoidf_(gr[3 * H + j] + bias2[3 * H + j]);
  gr[j] = i; gr[H + j] = f; gr[2 * H + j] = g; gr[3 * H + j] = o;
  const float cp = c_prev ? c_prev[idx] : 0.f;
  const float c = f * cp + i * g;
  const float h = o * tanhf(c);
  c_out[idx] = c;
  h_out[(long)b * ldh + j] = h;
  if (h_out2) h_out2[(long)b * ldh2 + j] = h;
}

// ---- cell backward of a step whose hidden gradient is given (last step of the forward direction,
//      the single step of the reverse direction).  dc (if not null) receives the gradient for the
//      cell state before the step.
struct CellBwdInst {
  const float *gates, *c_prev, *c, *dh;
  float *dc, *dgates;
  int lddh, add_dc;     // add_dc: dc holds the cell gradient arriving from the following step
};
struct CellBwdGroup { CellBwdInst in[2 * kMaxInst]; };
__global__ void lstm_cell_backward_kernel(CellBwdGroup grp, int B, int H) {
  const CellBwdInst& g = grp.in[blockIdx.z];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, j = idx % H;
  const float* gr = g.gates + (long)b * 4 * H;
  const float i = gr[j], f = gr[H + j], gg = gr[2 * H + j], o = gr[3 * H + j];
  const float tc = tanhf(g.c[idx]);
  const float dhv = g.dh[(long)b * g.lddh + j];
  const float dcv = (g.add_dc ? g.dc[idx] : 0.f) + dhv * o * (1.f - tc * tc);
  const float cp = g.c_prev ? g.c_prev[idx] : 0.f;
  float* dg = g.dgates + (long)b * 4 * H;
  dg[j] = dcv * gg * i * (1.f - i);
  dg[H + j] = dcv * cp * f * (1.f - f);
  dg[2 * H + j] = dcv * i * (1.f - gg * gg);
  dg[3 * H + j] = dhv * tc * o * (1.f - o);
  if (g.dc) g.dc[idx] = dcv * f;
}

// ---- one backward time step: dh_{t-1} = dgates_t * W_hh on the matrix cores, then, in the
// epilogue, the cell's backward for step t-1 on the same (row, unit) elements:
//   dc_{t-1} = dc + dh_{t-1} * o * (1 - tanh(c_{t-1})^2), the four gate gradients of step t-1,
//   and dc <- dc_{t-1} * f for the step before.
// A workgroup owns 16 batch rows x 16 hidden units (small tiles, many workgroups: the step is bound by
// how fast one CU can pull its operands); its four waves each reduce one gate's H columns of dgates_t
// on v_mfma_f32_16x16x4_f32, partial blocks are summed in LDS in wave order.  Every (row, unit) element
// is owned by exactly one thread of the grid, so dc is updated in place and dh never goes to memory.
struct StepBwdInst {
  const float *dgates_t, *w_hh, *gates_p, *c_pp, *c_p;   // gates_p: activated gates of step t-1; c_pp: cell before it (NULL = 0)
  float *dc, *dgates_p;
};
struct StepBwdGroup { StepBwdInst in[kMaxInst]; };
typedef float f32x4 __attribute__((ext_vector_type(4)));
static int step_bwd_lda(int H) { return 4 * H + 4; }       // row stride of the dgates image: 4 mod 32 banks, 16-byte rows
static long step_bwd_lds_bytes(int H) { return 4L * (16L * step_bwd_lda(H) + 4L * H * 17 + 4 * 16 * 17); }

template <int NT>
__global__ __launch_bounds__(NT) void lstm_step_backward_kernel(StepBwdGroup grp, int B, int H) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const StepBwdInst& g = grp.in[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u0 = blockIdx.x * 16, row0 = blockIdx.y * 16;
  const int G4 = 4 * H, lda = G4 + 4;
  float* Dg = lds;                        // [16][4H+4]  dgates_t rows
  float* Wt = Dg + 16 * lda;              // [4H][17]    W_hh[k][u0 + j]
  float* part = Wt + G4 * 17;             // [4][16][17]
  {
    // 16-byte loads, 2048 / NT + 2048 / NT per thread for H = 128, all issued before the LDS writes
    const int a4 = 16 * H;                // float4 count of the dgates tile (16 rows x 4H)
    const int n4 = G4 * 4;                // W_hh[k][u0 .. u0+15]: 4 float4 per row k
    constexpr int PER = 2048 / NT;
    for (int base = 0; base < a4; base += PER * NT) {
      float4 v[PER], wv[PER];
#pragma unroll
      for (int it = 0; it < PER; ++it) {
        const int i4 = base + it * NT + tid;
        const int rr = i4 / H, c4 = i4 - rr * H;
        v[it] = (i4 < a4 && row0 + rr < B) ? reinterpret_cast<const float4*>(g.dgates_t + (long)(row0 + rr) * G4)[c4]
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
        const int k = i4 >> 2, w4 = i4 & 3;
        wv[it] = (i4 < n4 && u0 + 4 * w4 + 3 < H) ? reinterpret_cast<const float4*>(g.w_hh + (long)k * H + u0)[w4]
                                                    : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int it = 0; it < PER; ++it) {
        const int i4 = base + it * NT + tid;
        const int rr = i4 / H, c4 = i4 - rr * H;
        if (i4 < a4) *reinterpret_cast<float4*>(Dg + rr * lda + 4 * c4) = v[it];
        const int k = i4 >> 2, w4 = i4 & 3;
        if (i4 < n4) {
          float* p = Wt + k * 17 + 4 * w4;
          p[0] = wv[it].x; p[1] = wv[it].y; p[2] = wv[it].z; p[3] = wv[it].w;
        }
      }
    }
  }
  __syncthreads();
  if (wave < 4) {
    // wave `wave` reduces gate `wave`: k in [wave*H, (wave+1)*H), two accumulation chains
    const int i = lane & 15, q = lane >> 4;
    const float* ap = Dg + i * lda + wave * H + q;
    const float* bp = Wt + (wave * H + q) * 17 + i;
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < H; k += 8) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[k], bp[k * 17], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[k + 4], bp[(k + 4) * 17], c1, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) part[(wave * 16 + 4 * q + e) * 17 + i] = c0[e] + c1[e];
  }
  __syncthreads();
  if (wave < 4) {
    const int rr = tid >> 4, j = tid & 15;
    const int b = row0 + rr, u = u0 + j;
    if (b < B && u < H) {
      const float dhv = (part[(0 * 16 + rr) * 17 + j] + part[(1 * 16 + rr) * 17 + j]) +
                        (part[(2 * 16 + rr) * 17 + j] + part[(3 * 16 + rr) * 17 + j]);
      const float* gr = g.gates_p + (long)b * G4;
      const float gi = gr[u], gf = gr[H + u], gg = gr[2 * H + u], go = gr[3 * H + u];
      const float tc = tanhf(g.c_p[(long)b * H + u]);
      const float dcv = g.dc[(long)b * H + u] + dhv * go * (1.f - tc * tc);
      const float cp = g.c_pp ? g.c_pp[(long)b * H + u] : 0.f;
      float* dg = g.dgates_p + (long)b * G4;
      dg[u] = dcv * gg * gi * (1.f - gi);
      dg[H + u] = dcv * cp * gf * (1.f - gf);
      dg[2 * H + u] = dcv * gi * (1.f - gg * gg);
      dg[3 * H + u] = dhv * tc * go * (1.f - go);
      g.dc[(long)b * H + u] = dcv * gf;
    }
  }
}

// ---- db1[col] += colsum(x), db2[col] += colsum(x)   (bias_ih and bias_hh share a gradient)
struct ColsumInst { const float* x; float *db1, *db2; int M; };
struct ColsumGroup { ColsumInst in[2 * kMaxInst]; };
__global__ __launch_bounds__(256) void colsum2_kernel(ColsumGroup grp, int ld, int N, int rows_per_block) {
  __shared__ float red[4][64];
  const ColsumInst& g = grp.in[blockIdx.z];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(g.M, r0 + rows_per_block);
  if (r0 >= g.M) return;
  float s = 0.f;
  if (col < N) {
    // 16 rows per thread and pass, all requested before the first add (the loop is latency-bound otherwise)
    for (int rb = r0 + ty; rb < r1; rb += 64) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int r = rb + 4 * u;
        v[u] = r < r1 ? g.x[(long)r * ld + col] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && col < N) {
    s = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
    unsafeAtomicAdd(&g.db1[col], s);
    if (g.db2) unsafeAtomicAdd(&g.db2[col], s);
  }
}

static int set_step_attrs(int H) {
  HHRL_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(lstm_step_forward_kernel<256>), (int)step_fwd_lds_bytes(H), "lstm"));
  HHRL_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(lstm_step_forward_kernel<512>), (int)step_fwd_lds_bytes(H), "lstm"));
  HHRL_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(lstm_step_backward_kernel<256>), (int)step_bwd_lds_bytes(H), "lstm"));
  HHRL_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(lstm_step_backward_kernel<512>), (int)step_bwd_lds_bytes(H), "lstm"));
  return HHRL_OK;
}

// The fused step kernels hold their tiles in LDS and assume H % 32 == 0.
static bool fused_steps_fit(const hhrl_lstm_desc* d, int n, int H) {
  if (H % 32 != 0 || H > 128) return false;
  for (int i = 0; i < n; ++i)
    if (d[i].o > kMaxObsPad) return false;
  return true;
}

static int check_group(const hhrl_lstm_desc* d, int n, int B, int T, int H, bool backward) {
  HHRL_REQUIRE(d != nullptr && n >= 1 && n <= kMaxInst, "lstm: group of %d instances (1..%d supported)", n, kMaxInst);
  HHRL_REQUIRE(B > 0 && T > 0 && H > 0, "lstm: bad shape B=%d T=%d H=%d", B, T, H);
  for (int i = 0; i < n; ++i) {
    HHRL_REQUIRE(d[i].ws && d[i].params, "lstm: instance %d: null pointer", i);
    HHRL_REQUIRE(d[i].o > 0, "lstm: instance %d: observation width %d", i, d[i].o);
    for (int k = 0; k < 10; ++k) HHRL_REQUIRE(d[i].params[k], "lstm: instance %d: params[%d] is null", i, k);
    if (!backward) {
      HHRL_REQUIRE(d[i].obs && d[i].out, "lstm_forward: instance %d: null pointer", i);
      HHRL_REQUIRE(d[i].ldo >= H && d[i].ldobs >= T * d[i].o, "lstm_forward: instance %d: ldo=%d ldobs=%d too small", i, d[i].ldo, d[i].ldobs);
    } else {
      HHRL_REQUIRE(d[i].d_out && d[i].grads && d[i].lddo >= H, "lstm_backward: instance %d: null pointer or lddo=%d < H", i, d[i].lddo);
      for (int k = 0; k < 10; ++k) HHRL_REQUIRE(d[i].grads[k], "lstm_backward: instance %d: grads[%d] is null", i, k);
    }
  }
  return HHRL_OK;
}

}  // namespace hhrl

using namespace hhrl;

extern "C" {

long hhrl_lstm_workspace_floats(int B, int T, int o, int H) {
  if (B <= 0 || T <= 0 || o <= 0 || H <= 0) return 0;
  return carve(nullptr, B, T, o, H).total;
}

int hhrl_lstm_forward_group(const hhrl_lstm_desc* d, int n, int B, int T, int H, void* stream) {
  HHRL_TRY(check_group(d, n, B, T, H, false));
  if (fused_steps_fit(d, n, H)) HHRL_TRY(set_step_attrs(H));
  hipStream_t st = as_stream(stream);
  const int G = 4 * H;
  const long BH = (long)B * H;
  LstmWs w[kMaxInst];
  int omax = 0;
  PrepGroup pg;
  for (int i = 0; i < n; ++i) {
    w[i] = carve(d[i].ws, B, T, d[i].o, H);
    omax = d[i].o > omax ? d[i].o : omax;
    pg.in[i] = PrepInst{d[i].obs, w[i].x_tm, w[i].hbuf, w[i].cbuf, d[i].ldobs, d[i].o};
  }
  {
    const long work = (long)B * T * omax > BH ? (long)B * T * omax : BH;
    ProfScope ps("lstm_prep_kernel", st, 0.0, 8.0 * B * T * omax * n);
    hipLaunchKernelGGL(lstm_prep_kernel, dim3(ceil_div(work, 256), 1, n), dim3(256), 0, st, pg, B, T, H);
    HHRL_TRY(launch_status("lstm_prep_kernel"));
  }
  const bool fused = fused_steps_fit(d, n, H);
  if (!fused) {
    for (int i = 0; i < n; ++i) {
      const float* const* p = d[i].params;
      const int o = d[i].o;
      HHRL_TRY(gemm(w[i].x_tm, o, 1, p[0], 1, o, w[i].gates, G, T * B, G, o, p[2], HHRL_ACT_NONE, 0, st));
      HHRL_TRY(gemm(w[i].x_tm + (long)(T - 1) * B * o, o, 1, p[4], 1, o, w[i].gates_r, G, B, G, o, p[6], HHRL_ACT_NONE, 0, st));
      for (int t = 0; t < T; ++t) {
        float* gt = w[i].gates + (long)t * B * G;
        if (t > 0) HHRL_TRY(gemm(w[i].hbuf + (long)t * BH, H, 1, p[1], 1, H, gt, G, B, G, H, nullptr, HHRL_ACT_NONE, 1, st));
        hipLaunchKernelGGL(lstm_cell_forward_kernel, dim3(ceil_div(BH, 256)), dim3(256), 0, st, gt, p[3],
                           t > 0 ? (const float*)(w[i].cbuf + (long)t * BH) : (const float*)nullptr,
                           w[i].cbuf + (long)(t + 1) * BH, w[i].hbuf + (long)(t + 1) * BH, H,
                           t == T - 1 ? w[i].cat : (float*)nullptr, 2 * H, B, H);
        HHRL_TRY(launch_status("lstm_cell_forward_kernel"));
      }
      hipLaunchKernelGGL(lstm_cell_forward_kernel, dim3(ceil_div(BH, 256)), dim3(256), 0, st, w[i].gates_r, p[7],
                         (const float*)nullptr, w[i].c_r, w[i].cat + H, 2 * H, (float*)nullptr, 0, B, H);
      HHRL_TRY(launch_status("lstm_cell_forward_kernel(reverse)"));
    }
  }
  const dim3 sgrid_base(ceil_div(H, 8), ceil_div(B, 32));
  for (int t = 0; t < T && fused; ++t) {
    // the first launch also takes the single step of every reverse direction (zero state, input x[T-1])
    StepFwdGroup sg;
    int m = 0;
    const bool last = t == T - 1;
    for (int i = 0; i < n; ++i) {
      const float* const* p = d[i].params;
      StepFwdInst s;
      s.x = w[i].x_tm + (long)t * B * d[i].o;
      s.w_ih = p[0]; s.w_hh = p[1]; s.b_ih = p[2]; s.b_hh = p[3];
      s.h_prev = t > 0 ? w[i].hbuf + (long)t * BH : nullptr;
      s.c_prev = t > 0 ? w[i].cbuf + (long)t * BH : nullptr;
      s.gates = w[i].gates + (long)t * B * G;
      s.c_out = w[i].cbuf + (long)(t + 1) * BH;
      s.h_out = w[i].hbuf + (long)(t + 1) * BH;
      s.ldh = H;
      s.h_out2 = last ? w[i].cat : nullptr;
      s.ldh2 = 2 * H;
      s.o = d[i].o;
      sg.in[m++] = s;
    }
    int launched = 0;
    auto launch = [&](int count) -> int {
      ProfScope ps("lstm_step_forward_kernel", st, 2.0 * B * G * (double)(t > 0 ? H + omax : omax) * count,
                   4.0 * count * (B * (2.0 * G + 4.0 * H) + (double)G * H));
      if (step_threads() == 512)
        hipLaunchKernelGGL(lstm_step_forward_kernel<512>, dim3(sgrid_base.x, sgrid_base.y, count), dim3(512), step_fwd_lds_bytes(H), st,
                           sg, B, H);
      else
        hipLaunchKernelGGL(lstm_step_forward_kernel<256>, dim3(sgrid_base.x, sgrid_base.y, count), dim3(256), step_fwd_lds_bytes(H), st,
                           sg, B, H);
      launched += count;
      return launch_status("lstm_step_forward_kernel");
    };
    if (t == 0) {
      if (2 * n > kMaxInst) {          // too many for one table: forward steps now, reverse steps in a second launch
        HHRL_TRY(launch(m));
        m = 0;
      }
      for (int i = 0; i < n; ++i) {
        const float* const* p = d[i].params;
        StepFwdInst s;
        s.x = w[i].x_tm + (long)(T - 1) * B * d[i].o;
        s.w_ih = p[4]; s.w_hh = p[5]; s.b_ih = p[6]; s.b_hh = p[7];
        s.h_prev = nullptr; s.c_prev = nullptr;
        s.gates = w[i].gates_r;
        s.c_out = w[i].c_r;
        s.h_out = w[i].cat + H;
        s.ldh = 2 * H;
        s.h_out2 = nullptr;
        s.ldh2 = 0;
        s.o = d[i].o;
        sg.in[m++] = s;
      }
    }
    HHRL_TRY(launch(m));
  }
  // out = [h_T | h_reverse] * fc_w^T + fc_b
  GemmArgs fc[kMaxInst];
  for (int i = 0; i < n; ++i)
    fc[i] = gemm_args(w[i].cat, 2 * H, 1, d[i].params[8], 1, 2 * H, d[i].out, d[i].ldo, B, H, 2 * H, d[i].params[9],
                      HHRL_ACT_NONE, 0);
  return gemm_group(fc, n, st);
}

int hhrl_lstm_backward_group(const hhrl_lstm_desc* d, int n, int B, int T, int H, void* stream) {
  HHRL_TRY(check_group(d, n, B, T, H, true));
  const bool fused = fused_steps_fit(d, n, H);
  if (fused) HHRL_TRY(set_step_attrs(H));
  hipStream_t st = as_stream(stream);
  const int G = 4 * H;
  const long BH = (long)B * H;
  LstmWs w[kMaxInst];
  for (int i = 0; i < n; ++i) w[i] = carve(d[i].ws, B, T, d[i].o, H);

  // fc: out = cat * fc_w^T + fc_b   ->  db, dcat = d_out * fc_w, dW += d_out^T cat
  {
    ColsumGroup cg;
    GemmArgs dx[kMaxInst], dw[kMaxInst];
    for (int i = 0; i < n; ++i) {
      cg.in[i] = ColsumInst{d[i].d_out, d[i].grads[9], nullptr, B};
      dx[i] = gemm_args(d[i].d_out, d[i].lddo, 1, d[i].params[8], 2 * H, 1, w[i].dcat, 2 * H, B, 2 * H, H, nullptr, HHRL_ACT_NONE, 0);
      dw[i] = gemm_args(d[i].d_out, 1, d[i].lddo, w[i].cat, 2 * H, 1, d[i].grads[8], 2 * H, H, 2 * H, B, nullptr, HHRL_ACT_NONE, 2);
    }
    // all instances of a group share lddo only by convention: the column sums take it per launch
    for (int i = 1; i < n; ++i)
      HHRL_REQUIRE(d[i].lddo == d[0].lddo, "lstm_backward: instances of a group must share lddo (%d vs %d)", d[i].lddo, d[0].lddo);
    {
      ProfScope ps("colsum2_kernel", st, 0.0, 4.0 * B * H * n);
      hipLaunchKernelGGL(colsum2_kernel, dim3(ceil_div(H, 64), ceil_div(B, 64), n), dim3(256), 0, st, cg, d[0].lddo, H, 64);
      HHRL_TRY(launch_status("colsum2_kernel(fc)"));
    }
    HHRL_TRY(gemm_group(dx, n, st));
    HHRL_TRY(gemm_group(dw, n, st));
  }
  // cells whose hidden gradient comes from the fc layer: last forward step and the reverse step
  {
    CellBwdGroup cg;
    for (int i = 0; i < n; ++i) {
      cg.in[2 * i] = CellBwdInst{w[i].gates + (long)(T - 1) * B * G, T > 1 ? w[i].cbuf + (long)(T - 1) * BH : nullptr,
                                 w[i].cbuf + (long)T * BH, w[i].dcat, w[i].dc, w[i].dgates + (long)(T - 1) * B * G, 2 * H, 0};
      cg.in[2 * i + 1] = CellBwdInst{w[i].gates_r, nullptr, w[i].c_r, w[i].dcat + H, nullptr, w[i].dgates_r, 2 * H, 0};
    }
    ProfScope ps("lstm_cell_backward_kernel", st, 0.0, 48.0 * BH * 2 * n);
    hipLaunchKernelGGL(lstm_cell_backward_kernel, dim3(ceil_div(BH, 256), 1, 2 * n), dim3(256), 0, st, cg, B, H);
    HHRL_TRY(launch_status("lstm_cell_backward_kernel"));
  }
  // steps T-1 .. 1: dh_{t-1} = dgates_t W_hh fused with the cell backward of step t-1
  for (int t = T - 1; t >= 1 && !fused; --t) {
    // generic hidden size: dh_{t-1} = dgates_t W_hh as a GEMM, then the cell backward of step t-1
    CellBwdGroup cg;
    for (int i = 0; i < n; ++i) {
      HHRL_TRY(gemm(w[i].dgates + (long)t * B * G, G, 1, d[i].params[1], H, 1, w[i].dh, H, B, H, G, nullptr, HHRL_ACT_NONE, 0, st));
      cg.in[i] = CellBwdInst{w[i].gates + (long)(t - 1) * B * G, t > 1 ? w[i].cbuf + (long)(t - 1) * BH : nullptr,
                             w[i].cbuf + (long)t * BH, w[i].dh, w[i].dc, w[i].dgates + (long)(t - 1) * B * G, H, 1};
    }
    hipLaunchKernelGGL(lstm_cell_backward_kernel, dim3(ceil_div(BH, 256), 1, n), dim3(256), 0, st, cg, B, H);
    HHRL_TRY(launch_status("lstm_cell_backward_kernel"));
  }
  for (int t = T - 1; t >= 1 && fused; --t) {
    StepBwdGroup sg;
    for (int i = 0; i < n; ++i)
      sg.in[i] = StepBwdInst{w[i].dgates + (long)t * B * G, d[i].params[1], w[i].gates + (long)(t - 1) * B * G,
                             t > 1 ? w[i].cbuf + (long)(t - 1) * BH : nullptr, w[i].cbuf + (long)t * BH, w[i].dc,
                             w[i].dgates + (long)(t - 1) * B * G};
    ProfScope ps("lstm_step_backward_kernel", st, 2.0 * B * G * H * n, 4.0 * n * (B * (3.0 * G + 4.0 * H) + (double)G * H));
    if (step_threads() == 512)
      hipLaunchKernelGGL(lstm_step_backward_kernel<512>, dim3(ceil_div(H, 16), ceil_div(B, 16), n), dim3(512), step_bwd_lds_bytes(H), st,
                       sg, B, H);
    else
      hipLaunchKernelGGL(lstm_step_backward_kernel<256>, dim3(ceil_div(H, 16), ceil_div(B, 16), n), dim3(256), step_bwd_lds_bytes(H), st,
                       sg, B, H);
    HHRL_TRY(launch_status("lstm_step_backward_kernel"));
  }
  // weight gradients of the whole window: rows (t,b) of dgates against x_tm / previous hidden state
  {
    GemmArgs dw[3 * kMaxInst];
    ColsumGroup cg;
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const int o = d[i].o;
      dw[m++] = gemm_args(w[i].dgates, 1, G, w[i].x_tm, o, 1, d[i].grads[0], o, G, o, T * B, nullptr, HHRL_ACT_NONE, 2);
      dw[m++] = gemm_args(w[i].dgates, 1, G, w[i].hbuf, H, 1, d[i].grads[1], H, G, H, T * B, nullptr, HHRL_ACT_NONE, 2);
      dw[m++] = gemm_args(w[i].dgates_r, 1, G, w[i].x_tm + (long)(T - 1) * B * o, o, 1, d[i].grads[4], o, G, o, B, nullptr,
                          HHRL_ACT_NONE, 2);
      cg.in[2 * i] = ColsumInst{w[i].dgates, d[i].grads[2], d[i].grads[3], T * B};
      cg.in[2 * i + 1] = ColsumInst{w[i].dgates_r, d[i].grads[6], d[i].grads[7], B};
    }
    for (int base = 0; base < m; base += kGemmGroupMax)
      HHRL_TRY(gemm_group(dw + base, m - base < kGemmGroupMax ? m - base : kGemmGroupMax, st));
    ProfScope ps("colsum2_kernel", st, 0.0, 4.0 * (T + 1) * B * G * n);
    hipLaunchKernelGGL(colsum2_kernel, dim3(ceil_div(G, 64), ceil_div(T * B, 64), 2 * n), dim3(256), 0, st, cg, G, G, 64);
    HHRL_TRY(launch_status("colsum2_kernel"));
  }
  return HHRL_OK;
}

int hhrl_lstm_forward(const float* obs, int ldobs, const float* const* params, float* out, int ldo, float* ws, int B,
                      int T, int o, int H, void* stream) {
  hhrl_lstm_desc d;
  d.obs = obs; d.ldobs = ldobs; d.o = o;
  d.params = params; d.grads = nullptr;
  d.out = out; d.ldo = ldo;
  d.d_out = nullptr; d.lddo = 0;
  d.ws = ws;
  return hhrl_lstm_forward_group(&d, 1, B, T, H, stream);
}

int hhrl_lstm_backward(const float* const* params, const float* d_out, int lddo, float* const* grads, float* ws,
                       int B, int T, int o, int H, void* stream) {
  hhrl_lstm_desc d;
  d.obs = nullptr; d.ldobs = 0; d.o = o;
  d.params = params; d.grads = grads;
  d.out = nullptr; d.ldo = 0;
  d.d_out = d_out; d.lddo = lddo;
  d.ws = ws;
  return hhrl_lstm_backward_group(&d, 1, B, T, H, stream);
}

}  // extern "C"
