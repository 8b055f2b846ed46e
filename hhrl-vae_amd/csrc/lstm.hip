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
  float* dwslab;          // [win_splits][4H*H + 4H*o + 4H] partial window weight gradients (long windows x batches only)
  long total;
};

// The window's weight gradients as a reduction over the rows (t, b) split into this many slabs, one workgroup row
// each (lstm_window_dw_kernel); 0 = the grouped small-product kernels serve the size.
constexpr int kWinKc = 32;                    // rows (t, b) per staged chunk
constexpr long kWinMinRows = 2048;
static int win_splits(long B, long T, long o, long H) {
  const long K = T * B;
  if (H != 128 || o > 32 || K < kWinMinRows) return 0;
  const long s = (K + 4 * kWinKc - 1) / (4 * kWinKc);      // at least four chunks per slab
  return (int)(s < 64 ? s : 64);
}
// slabs of the reverse direction's B rows in the same launch
static int win_splits_rev(long B) {
  const long s = (B + 4 * kWinKc - 1) / (4 * kWinKc);
  return (int)(s < 8 ? s : 8);
}

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
  w.dwslab = take(win_splits(B, T, o, H) > 0 ? (long)(win_splits(B, T, o, H) + win_splits_rev(B)) * (4 * H * H + 4 * H * o + 4 * H) : 0);
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
  // what the gate epilogue reads from memory (biases, the cell state before the step) is requested now,
  // so that its latency passes under the staging and the product instead of after the last barrier
  const int e_rr = tid >> 3, e_j = tid & 7;
  const int e_b = row0 + e_rr, e_u = u0 + e_j;
  const bool e_on = wave < 4 && e_b < B && e_u < H;
  float e_bias[4] = {0.f, 0.f, 0.f, 0.f}, e_cp = 0.f;
  if (e_on) {
#pragma unroll
    for (int q = 0; q < 4; ++q) e_bias[q] = g.b_ih[q * H + e_u] + g.b_hh[q * H + e_u];
    if (g.c_prev) e_cp = g.c_prev[(long)e_b * H + e_u];
  }
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
    const int rr = e_rr, j = e_j;
    const int b = e_b, u = e_u;
    if (e_on) {
      float pre[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = 8 * q + j;
        pre[q] = (part[(0 * 32 + rr) * 33 + col] + part[(1 * 32 + rr) * 33 + col]) +
                 (part[(2 * 32 + rr) * 33 + col] + part[(3 * 32 + rr) * 33 + col]);
        pre[q] += e_bias[q];
      }
      float* gr = g.gates + (long)b * 4 * H;
      const float gi = sigmoidf_(pre[0]);
      const float gf = sigmoidf_(pre[1]);
      const float gg = tanhf(pre[2]);
      const float go = sigmoidf_(pre[3]);
      gr[u] = gi; gr[H + u] = gf; gr[2 * H + u] = gg; gr[3 * H + u] = go;
      const float cp = e_cp;
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
  // the cell epilogue's operands (gates and cell states of step t-1, the running cell gradient) are requested
  // now: their latency passes under the staging and the product
  const int e_rr = tid >> 4, e_j = tid & 15;
  const int e_b = row0 + e_rr, e_u = u0 + e_j;
  const bool e_on = wave < 4 && e_b < B && e_u < H;
  float e_g[4] = {0.f, 0.f, 0.f, 0.f}, e_cpv = 0.f, e_dc = 0.f, e_cpp = 0.f;
  if (e_on) {
    const float* gr = g.gates_p + (long)e_b * G4;
#pragma unroll
    for (int q = 0; q < 4; ++q) e_g[q] = gr[q * H + e_u];
    e_cpv = g.c_p[(long)e_b * H + e_u];
    e_dc = g.dc[(long)e_b * H + e_u];
    if (g.c_pp) e_cpp = g.c_pp[(long)e_b * H + e_u];
  }
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
    const int rr = e_rr, j = e_j;
    const int b = e_b, u = e_u;
    if (e_on) {
      const float dhv = (part[(0 * 16 + rr) * 17 + j] + part[(1 * 16 + rr) * 17 + j]) +
                        (part[(2 * 16 + rr) * 17 + j] + part[(3 * 16 + rr) * 17 + j]);
      const float gi = e_g[0], gf = e_g[1], gg = e_g[2], go = e_g[3];
      const float tc = tanhf(e_cpv);
      const float dcv = e_dc + dhv * go * (1.f - tc * tc);
      const float cp = e_cpp;
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

// =====================================================================================================
// The whole recurrence of a window in ONE launch (H = 128, per-step observation at most 32 wide).
//
// A workgroup (eight waves) owns four batch rows of one module and all 512 gate columns, so the hidden state
// never leaves the CU between time steps.  The products run on v_mfma_f32_4x4x1_f32, which computes sixteen
// independent 4x4 outer products per instruction: operand A holds the same four batch rows for every block
// (lane 4b+i = row i), operand B one gate column per lane, the result register e of lane l is
// (row e, column of lane l).  A lane therefore keeps ITS gate column of [W_hh | W_ih] in registers for the whole
// window (128 + OPAD values, loaded once), and a time step reads only the four rows [h_{t-1} | x_t] from LDS:
// no weight traffic per step.  Wave w owns hidden units 16w..16w+15 with their four gates (lane = 16*gate +
// unit), so the cell update needs a wave-local exchange only; one workgroup barrier per step publishes h_t.
// Batch rows beyond the first four are taken in further passes of the same workgroup (weights stay in
// registers); workgroups with blockIdx.y = 1 compute the single step of the reverse direction (zero state,
// input x[T-1]) on the vector ALU.
typedef float f32x4s __attribute__((ext_vector_type(4)));
// Diagnostic builds (-DHHRL_STAMPS): shader clock of workgroup (0,0,0), thread 0 at marked points of the
// one-launch recurrence; hhrl_debug_lstm_stamps() copies them out.
#ifdef HHRL_STAMPS
__device__ long long g_lstm_stamps[2][64];
__device__ int g_lstm_stamp_wg;     // first forward workgroup of the forward kernel's grid (the reverse ones come before it)
#define LSTM_STAMP(k, i) do { if (blockIdx.x == (k == 0 ? g_lstm_stamp_wg : 0) && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_lstm_stamps[k][i] = clock64(); } while (0)
#else
#define LSTM_STAMP(k, i) do { } while (0)
#endif
// Gate nonlinearities of the one-launch recurrence: the hardware exponential and reciprocal (about 1 ulp each)
// instead of the library routines, whose 25-45 instructions per call made the cell update as long as the products.
// sigmoid: relative error < 4e-7.  tanh: 1 - 2 / (e^{2x} + 1) has absolute error about 1e-7, which is a large
// RELATIVE error for small arguments, so below 0.25 the odd Taylor polynomial through x^9 takes over (next term < 3e-9).
__device__ __forceinline__ float sq_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float sq_tanh(float x) {
  const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f);
  const float x2 = x * x;
  const float small = x * (1.f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * (-0.053968254f + x2 * 0.021869488f))));
  return fabsf(x) < 0.25f ? small : big;
}
constexpr int SQ_H = 128, SQ_G = 512, SQ_XG = 72;      // SQ_XG: row stride of the wave-local gate exchange

struct SeqFwdInst {
  const float* obs;                                   // [B][ldobs], step t at columns [t*o, (t+1)*o)
  const float *w_ih, *w_hh, *b_ih, *b_hh;             // forward direction
  const float *w_ih_r, *b_ih_r, *b_hh_r;              // reverse direction
  float *x_tm, *gates, *cbuf, *hbuf, *cat, *gates_r, *c_r;
  int ldobs, o;
  int save;                                           // 0: nobody back-propagates this evaluation; only `cat` is written
};
// The launch is one row of workgroups; module m owns workgroups wg0[m] .. wg0[m+1]-1, the first n_rev[m] of them
// take the reverse direction's step.  Modules whose steps are not saved are cheaper per step and get fewer.
struct SeqFwdGroup {
  SeqFwdInst in[kMaxInst];
  short wg0[kMaxInst + 1], n_rev[kMaxInst];
  int n;
};

constexpr int SQ_SLAB = 512 * 33 + 64;                 // one 32-column slab of a [512][.] weight matrix, row stride 33 (+16 per gate)
template <int OPAD>
__global__ __launch_bounds__(512) void lstm_seq_forward_kernel(SeqFwdGroup grp, int B, int T, int chunks) {
  constexpr int H = SQ_H, G = SQ_G, LD = H + 4;           // LD * 4 bytes is a multiple of 16
  // dynamic LDS: the weight slab of the prologue, then the window's observations [T][4][OPAD] of the four rows
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  float* xall = dyn + SQ_SLAB;
  __shared__ __attribute__((aligned(16))) float hs[2][4][LD];     // h_{t-1} / h_t of the four rows
  __shared__ float xg[8][4][SQ_XG];
  int mod = 0;
  while (mod + 1 < grp.n && (int)blockIdx.x >= grp.wg0[mod + 1]) ++mod;
  const SeqFwdInst& g = grp.in[mod];
  const int wg = blockIdx.x - grp.wg0[mod], gx_all = grp.wg0[mod + 1] - grp.wg0[mod], gx_r = grp.n_rev[mod];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int o = g.o;
  const long BH = (long)B * H;
  const bool save = g.save != 0;

  if (wg < gx_r) {
    // ---- reverse direction: one step from the zero state on x[T-1]; thread = gate column.  These workgroups come
    // first in dispatch order and are few (each takes several groups of rows): they are done within microseconds.
    float (*pre_r)[G + 4] = reinterpret_cast<float (*)[G + 4]>(dyn);
    float wr[OPAD];
#pragma unroll
    for (int c = 0; c < OPAD; ++c) wr[c] = c < o ? g.w_ih_r[(long)tid * o + c] : 0.f;
    const float bias = g.b_ih_r[tid] + g.b_hh_r[tid];
    const int xr = tid / OPAD, xc = tid - xr * OPAD;
    const int row = tid >> 7, unit = tid & 127;
    for (int chunk = wg; chunk < chunks; chunk += gx_r) {
      const int r0 = chunk * 4;
      if (tid < 4 * OPAD)
        xall[xr * OPAD + xc] = (xc < o && r0 + xr < B) ? g.obs[(long)(r0 + xr) * g.ldobs + (long)(T - 1) * o + xc] : 0.f;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float s = bias;
#pragma unroll
        for (int c = 0; c < OPAD; ++c) s += xall[i * OPAD + c] * wr[c];
        pre_r[i][tid] = s;
      }
      __syncthreads();
      if (r0 + row < B) {
        const float gi = sigmoidf_(pre_r[row][unit]);
        const float gf = sigmoidf_(pre_r[row][H + unit]);
        const float gg = tanhf(pre_r[row][2 * H + unit]);
        const float go = sigmoidf_(pre_r[row][3 * H + unit]);
        const float c = gi * gg;                       // the cell state before the step is zero
        if (save) {
          float* gr = g.gates_r + (long)(r0 + row) * G;
          gr[unit] = gi; gr[H + unit] = gf; gr[2 * H + unit] = gg; gr[3 * H + unit] = go;
          g.c_r[(long)(r0 + row) * H + unit] = c;
        }
        g.cat[(long)(r0 + row) * 2 * H + H + unit] = go * tanhf(c);
      }
      __syncthreads();
    }
    return;
  }
  const int bx = wg - gx_r, gx_f = gx_all - gx_r;

  LSTM_STAMP(0, 0);
  // ---- forward direction.  Matrix role: lane = (gate, unit of the wave); cell role: lane = (row, unit of the wave)
  const int mg = lane >> 4, mu = 16 * wave + (lane & 15);
  const int wrow = mg * H + mu;                            // this lane's row of W_ih / W_hh
  // The window's observations of four rows: into LDS once, and out again time-major (the weight-gradient products
  // read them that way); nothing is loaded from memory inside the time loop.
  auto stage_x = [&](int r0) {
    for (int base = 0; base < T * 4 * OPAD; base += 4 * 512) {     // four loads in flight per thread
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = base + u * 512 + tid;
        const int xc = e % OPAD, rt = e / OPAD, xr = rt & 3, t = rt >> 2;
        const bool in = e < T * 4 * OPAD && xc < o && r0 + xr < B;
        v[u] = in ? g.obs[(long)(r0 + xr) * g.ldobs + (long)t * o + xc] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = base + u * 512 + tid;
        const int xc = e % OPAD, rt = e / OPAD, xr = rt & 3, t = rt >> 2;
        if (e < T * 4 * OPAD) {
          xall[e] = v[u];
          if (save && xc < o && r0 + xr < B) g.x_tm[((long)t * B + r0 + xr) * o + xc] = v[u];
        }
      }
    }
  };
  float wh[H], wi[OPAD];
  {
    // A lane needs one whole ROW of the weight matrices.  Read row by row from memory that is 64 cache lines per
    // instruction; instead the workgroup copies the matrices through LDS in slabs of 32 columns (coalesced: eight
    // threads per 128-byte run) and every lane picks its row out of the slab.  The next slab's loads are issued
    // before the current one is written.
    const int lrow = wrow * 33 + 16 * mg;                  // 16 per gate: the four gates of a lane group fall on different banks
    float4 nx[8];
    auto request = [&](int slab) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int f = it * 512 + tid, row = f >> 3, c4 = f & 7;
        nx[it] = *reinterpret_cast<const float4*>(g.w_hh + (long)row * H + 32 * slab + 4 * c4);
      }
    };
    request(0);
    // W_ih: a lane's row is o contiguous floats and sixteen lanes read adjacent rows, so these go straight to registers
#pragma unroll
    for (int c = 0; c < OPAD; ++c) wi[c] = g.w_ih[(long)wrow * o + (c < o ? c : 0)];
#pragma unroll
    for (int c = 0; c < OPAD; ++c) wi[c] = c < o ? wi[c] : 0.f;
    // the first group's observations ride on the same memory round trip as the weights' first requests
    if (bx < chunks) stage_x(bx * 4);
    LSTM_STAMP(0, 8);
    // the barriers of this loop wait for LDS traffic only, so the next slab's loads stay in flight across them
#pragma unroll
    for (int slab = 0; slab < 4; ++slab) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int f = it * 512 + tid, row = f >> 3, c4 = f & 7;
        float* p = dyn + row * 33 + 16 * (row >> 7) + 4 * c4;
        p[0] = nx[it].x; p[1] = nx[it].y; p[2] = nx[it].z; p[3] = nx[it].w;
      }
      if (slab + 1 < 4) request(slab + 1);
      lds_barrier();
#pragma unroll
      for (int k = 0; k < 32; ++k) wh[32 * slab + k] = dyn[lrow + k];
      lds_barrier();
      if (slab == 0) LSTM_STAMP(0, 9);
    }
  }
  LSTM_STAMP(0, 10);
  const float bias = g.b_ih[wrow] + g.b_hh[wrow];
  const int crow = lane >> 4, cu = mu;                     // cell role
  const int arow = lane & 3;                               // operand A: lane 4b+i carries batch row i

  for (int chunk = bx; chunk < chunks; chunk += gx_f) {
    const int r0 = chunk * 4;
    const bool valid = r0 + crow < B;
    if (chunk != bx) stage_x(r0);                          // (the first group's were requested with the weights)
    if (valid && save) {                                   // the zero slot of the state buffers
      g.hbuf[(long)(r0 + crow) * H + cu] = 0.f;
      g.cbuf[(long)(r0 + crow) * H + cu] = 0.f;
    }
    float c = 0.f;
    // this thread's outputs of step 0; they advance by one time slice per step
    const int vrow = valid ? r0 + crow : 0;
    float* p_gate = g.gates + (long)vrow * G + cu;
    float* p_c = g.cbuf + BH + (long)vrow * H + cu;
    float* p_h = g.hbuf + BH + (long)vrow * H + cu;
    // Everything requested so far (weights, bias, observations) is waited for HERE: inside the loop the compiler
    // would otherwise wait for "all memory traffic" at the first use, which there includes the previous step's stores.
    LSTM_STAMP(0, 11);
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
    lds_barrier();
    LSTM_STAMP(0, 1);
    for (int t = 0; t < T; ++t) {
      const int cur = t & 1;
      f32x4s a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
      const float* hp = &hs[cur][arow][0];
      const float* xp = xall + (t * 4 + arow) * OPAD;
      constexpr int XQ = OPAD / 4;                         // 16-byte groups of the observation part
      auto group = [&](int q) -> float4 {
        return q < XQ ? *reinterpret_cast<const float4*>(xp + 4 * q) : *reinterpret_cast<const float4*>(hp + 4 * (q - XQ));
      };
      auto weight = [&](int k) -> float { return k < OPAD ? wi[k < OPAD ? k : 0] : wh[k < OPAD ? 0 : k - OPAD]; };
      if (t == 0) {                                        // zero state: the observation part only
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
          const float4 v = group(q);
          a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(v.x, wi[4 * q], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(v.y, wi[4 * q + 1], a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(v.z, wi[4 * q + 2], a2, 0, 0, 0);
          a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(v.w, wi[4 * q + 3], a3, 0, 0, 0);
        }
      } else {
        // Blocks of BG groups (4 * BG products = 32 * BG matrix-pipe cycles): the LDS reads of block n+1 are issued
        // before the products of block n, so that their latency passes under them.
        constexpr int BG = OPAD > 16 ? 2 : 4;               // groups per block (the wide form has fewer registers to spare)
        constexpr int NB = (XQ + H / 4) / BG;
        float4 va[BG], vb[BG];
#pragma unroll
        for (int i = 0; i < BG; ++i) va[i] = group(i);
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
          if (blk + 1 < NB) {
#pragma unroll
            for (int i = 0; i < BG; ++i) vb[i] = group(BG * (blk + 1) + i);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < BG; ++i) {
            const int k = 4 * BG * blk + 4 * i;
            a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(va[i].x, weight(k), a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(va[i].y, weight(k + 1), a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(va[i].z, weight(k + 2), a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(va[i].w, weight(k + 3), a3, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < BG; ++i) va[i] = vb[i];
        }
      }
      if (t == 1) LSTM_STAMP(0, 2);
      if (t == 2) LSTM_STAMP(0, 6);
#pragma unroll
      for (int i = 0; i < 4; ++i) xg[wave][i][lane] = ((a0[i] + a1[i]) + (a2[i] + a3[i])) + bias;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const float* pr = &xg[wave][crow][lane & 15];
      const float gi = sq_sigmoid(pr[0]);
      const float gf = sq_sigmoid(pr[16]);
      const float gg = sq_tanh(pr[32]);
      const float go = sq_sigmoid(pr[48]);
      c = gf * c + gi * gg;
      const float hv = go * sq_tanh(c);
      if (t == 1) LSTM_STAMP(0, 3);
      hs[cur ^ 1][crow][cu] = hv;
      if (valid) {
        if (save) {
          p_gate[0] = gi; p_gate[H] = gf; p_gate[2 * H] = gg; p_gate[3 * H] = go;
          *p_c = c;
          *p_h = hv;
          p_gate += (long)B * G;
          p_c += BH;
          p_h += BH;
        }
        if (t == T - 1) g.cat[(long)(r0 + crow) * 2 * H + cu] = hv;
      }
      if (t == 1) LSTM_STAMP(0, 4);
      lds_barrier();
      if (t == 1) LSTM_STAMP(0, 5);
      if (t == T - 1) LSTM_STAMP(0, 7);
    }
  }
}

// Back-propagation through the whole window in one launch, same ownership (four batch rows, all units).
// Matrix role: wave w reduces gate w/2's 128 columns of dgates_t for the 64 units of half w%2 (lane = unit),
// with its 128 values of W_hh in registers; the four partial results per (row, unit) are summed through LDS in
// gate order.  Cell role: thread = (row, unit) keeps the running cell gradient in a register.
struct SeqBwdInst {
  const float *w_hh, *gates, *cbuf, *dcat, *gates_r, *c_r;
  float *dgates, *dgates_r;
};
struct SeqBwdGroup { SeqBwdInst in[kMaxInst]; };

__global__ __launch_bounds__(512) void lstm_seq_backward_kernel(SeqBwdGroup grp, int B, int T, int chunks) {
  constexpr int H = SQ_H, G = SQ_G, LD = G + 4;
  __shared__ __attribute__((aligned(16))) float dg[4][LD];
  __shared__ float part[4][4][H + 1];
  const SeqBwdInst& g = grp.in[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = tid >> 7, unit = tid & 127;              // cell role
  const long BH = (long)B * H;

  if (blockIdx.y == 1) {
    // ---- the reverse direction's single cell: hidden gradient from the fc layer, zero state before the step
    for (int chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
      const int b = chunk * 4 + row;
      if (b >= B) continue;
      const float* gr = g.gates_r + (long)b * G;
      const float gi = gr[unit], gg = gr[2 * H + unit], go = gr[3 * H + unit];
      const float tc = tanhf(g.c_r[(long)b * H + unit]);
      const float dhv = g.dcat[(long)b * 2 * H + H + unit];
      const float dcv = dhv * go * (1.f - tc * tc);
      float* d = g.dgates_r + (long)b * G;
      d[unit] = dcv * gg * gi * (1.f - gi);
      d[H + unit] = 0.f;
      d[2 * H + unit] = dcv * gi * (1.f - gg * gg);
      d[3 * H + unit] = dhv * tc * go * (1.f - go);
    }
    return;
  }

  LSTM_STAMP(1, 0);
  const int ch = wave & 1, gq = wave >> 1;
  float wq[H];
#pragma unroll
  for (int c = 0; c < H; ++c) wq[c] = g.w_hh[((long)gq * H + c) * H + 64 * ch + lane];
  const int arow = lane & 3;

  for (int chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
    const int b = chunk * 4 + row;
    const bool valid = b < B;
    float dc = 0.f;
    // cell backward of step s with hidden gradient dhv; leaves dgates_s in LDS and in memory
    auto cell = [&](int s, float dhv, float gi, float gf, float gg, float go, float cs, float cp) {
      const float tc = sq_tanh(cs);
      const float dcv = dc + dhv * go * (1.f - tc * tc);
      const float d0 = dcv * gg * gi * (1.f - gi);
      const float d1 = dcv * cp * gf * (1.f - gf);
      const float d2 = dcv * gi * (1.f - gg * gg);
      const float d3 = dhv * tc * go * (1.f - go);
      dc = dcv * gf;
      dg[row][unit] = valid ? d0 : 0.f;
      dg[row][H + unit] = valid ? d1 : 0.f;
      dg[row][2 * H + unit] = valid ? d2 : 0.f;
      dg[row][3 * H + unit] = valid ? d3 : 0.f;
      if (valid) {
        float* d = g.dgates + ((long)s * B + b) * G;
        d[unit] = d0; d[H + unit] = d1; d[2 * H + unit] = d2; d[3 * H + unit] = d3;
      }
    };
    float n_g[4] = {0.f, 0.f, 0.f, 0.f}, n_c = 0.f, n_cp = 0.f;
    auto fetch = [&](int s) {                              // operands of the cell backward of step s
      if (!valid) return;
      const float* gr = g.gates + ((long)s * B + b) * G;
#pragma unroll
      for (int q = 0; q < 4; ++q) n_g[q] = gr[q * H + unit];
      n_c = g.cbuf[BH + ((long)s * B + b) * H + unit];
      n_cp = s > 0 ? g.cbuf[BH + ((long)(s - 1) * B + b) * H + unit] : 0.f;
    };
    fetch(T - 1);
    cell(T - 1, valid ? g.dcat[(long)b * 2 * H + unit] : 0.f, n_g[0], n_g[1], n_g[2], n_g[3], n_c, n_cp);
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): the weights; see the forward kernel
    lds_barrier();
    LSTM_STAMP(1, 1);
    for (int t = T - 1; t >= 1; --t) {
      fetch(t - 1);
      f32x4s a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
      const float* ap = &dg[arow][gq * H];
      {
        // blocks of four 16-byte groups, the reads of block n+1 issued before the products of block n
        float4 va[4], vb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) va[i] = *reinterpret_cast<const float4*>(ap + 4 * i);
#pragma unroll
        for (int blk = 0; blk < H / 16; ++blk) {
          if (blk + 1 < H / 16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) vb[i] = *reinterpret_cast<const float4*>(ap + 16 * (blk + 1) + 4 * i);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int k = 16 * blk + 4 * i;
            a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(va[i].x, wq[k], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(va[i].y, wq[k + 1], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(va[i].z, wq[k + 2], a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(va[i].w, wq[k + 3], a3, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i) va[i] = vb[i];
        }
      }
      if (t == T - 2) LSTM_STAMP(1, 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) part[gq][i][64 * ch + lane] = (a0[i] + a1[i]) + (a2[i] + a3[i]);
      lds_barrier();
      if (t == T - 2) LSTM_STAMP(1, 3);
      const float dhv = (part[0][row][unit] + part[1][row][unit]) + (part[2][row][unit] + part[3][row][unit]);
      cell(t - 1, dhv, n_g[0], n_g[1], n_g[2], n_g[3], n_c, n_cp);
      if (t == T - 2) LSTM_STAMP(1, 4);
      lds_barrier();
      if (t == T - 1) LSTM_STAMP(1, 6);
      if (t == T - 2) LSTM_STAMP(1, 5);
      if (t == 1) LSTM_STAMP(1, 7);
    }
  }
}

// The one-launch recurrence serves H = 128 with observations up to 32 wide and 16-byte aligned W_hh rows;
// HHRL_LSTM_SEQ=0 selects the per-step kernels instead (A/B timing).
constexpr int kSeqMaxT = 64;       // the forward kernel keeps the window's observations of its rows in LDS: T * 4 * 32 floats at most
static bool seq_fits(const hhrl_lstm_desc* d, int n, int H, int T) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("HHRL_LSTM_SEQ");
    on = (e && atoi(e) == 0) ? 0 : 1;
  }
  if (!on || H != SQ_H || T > kSeqMaxT) return false;
  for (int i = 0; i < n; ++i) {
    if (d[i].o > 32) return false;
    if (reinterpret_cast<uintptr_t>(d[i].params[1]) & 15) return false;
  }
  return true;
}
static int seq_grid_x(int chunks, int n) {
  // Forward workgroups of all modules plus the reverse-direction ones (a quarter as many) must fit ONE round of the
  // 256 CUs: a workgroup that has to wait for a CU starts when a forward workgroup ends, i.e. doubles the launch
  // (measured with 28 instead of 22 per module at nine modules, batch 1024: update 1.28 -> 1.43 ms).  Further groups
  // of rows are taken in passes of the same workgroup, whose weights stay in registers.
  int gx = 200 / n;
  if (gx < 1) gx = 1;
  return chunks < gx ? chunks : gx;
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


// ---- weight gradients of a whole window (H = 128, o <= 32): dW_hh [512][128] = dgates^T hprev, dW_ih [512][o] = dgates^T x,
// bias gradient = column sums of dgates, reductions over the T*B rows (t, b).  The output is small and the reduction
// long, so a workgroup takes 128 gate rows (blockIdx.x) x ALL 128 + o columns for one slab of the reduction range
// (blockIdx.y): its four waves own 32 gate rows each with five 32 x 32 accumulators (four column blocks of W_hh, one
// of W_ih), the operands of a chunk of 32 rows (t, b) are staged once per workgroup through LDS (two buffers, the next
// chunk's loads in flight during the products) and every staged dgates element feeds 160 products: 33 FLOP per byte
// fetched, against 8 for 32 x 32 output tiles.  The slabs' partial results go to the workspace and are summed in slab
// order by lstm_window_dw_reduce_kernel (the same bits on every run; no atomics).
struct WinDwInst {
  const float *dgates, *hprev, *x;            // forward direction: rows (t, b) of the window
  const float *dgates_r, *x_r;                // reverse direction: its single step on the last observation (hprev = 0)
  float *slab, *g_ih, *g_hh, *g_bih, *g_bhh, *g_ih_r, *g_bih_r, *g_bhh_r;
  int o;
};
struct WinDwGroup { WinDwInst in[kMaxInst]; };
constexpr int kWinLdsFloats = 2 * kWinKc * (129 + 129 + 33);
__global__ __launch_bounds__(256, 2) void lstm_window_dw_kernel(WinDwGroup grp, int K, int kslab, int splits, int Kr, int kslab_r) {
  extern __shared__ float wlds[];
  constexpr int H = 128, G = 512, KC = kWinKc;
  const WinDwInst& g = grp.in[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int row0 = blockIdx.x * 128, o = g.o;
  // slabs [0, splits): the forward direction's rows; the rest: the reverse direction's B rows (no recurrent part)
  const bool rev = (int)blockIdx.y >= splits;
  const int kbeg = rev ? ((int)blockIdx.y - splits) * kslab_r : blockIdx.y * kslab;
  const int kend = rev ? min(Kr, kbeg + kslab_r) : min(K, kbeg + kslab);
  const float* dg = rev ? g.dgates_r : g.dgates;
  const float* xin = rev ? g.x_r : g.x;
  float* As = wlds;                          // [2][KC][129]
  float* Bs = As + 2 * KC * 129;             // [2][KC][129]
  float* Xs = Bs + 2 * KC * 129;             // [2][KC][33]
  f32x16 acc[5];
#pragma unroll
  for (int c = 0; c < 5; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float rs = 0.f;
  float av[16], bv[16], xv[4];
  const int c128 = tid & 127, kk0 = tid >> 7;
  auto request = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int k = k0 + kk0 + 2 * u;
      const bool ok = k < kend;
      av[u] = ok ? dg[(long)k * G + row0 + c128] : 0.f;
      bv[u] = (ok && !rev) ? g.hprev[(long)k * H + c128] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u, kk = e >> 5, c = e & 31;
      xv[u] = (c < o && k0 + kk < kend) ? xin[(long)(k0 + kk) * o + c] : 0.f;
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      As[(buf * KC + kk0 + 2 * u) * 129 + c128] = av[u];
      Bs[(buf * KC + kk0 + 2 * u) * 129 + c128] = bv[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u;
      Xs[(buf * KC + (e >> 5)) * 33 + (e & 31)] = xv[u];
    }
  };
  if (kbeg < kend) {
    request(kbeg);
    commit(0);
  }
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += KC, buf ^= 1) {
    const bool more = k0 + KC < kend;
    if (more) request(k0 + KC);
    const float* a_ = As + buf * KC * 129 + 32 * wave + r;
    const float* b_ = Bs + buf * KC * 129 + r;
    const float* x_ = Xs + buf * KC * 33 + r;
#pragma unroll 4
    for (int k = 0; k < KC; k += 2) {
      const float a = a_[(k + h) * 129];
      rs += a;
      if (!rev) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b_[(k + h) * 129 + 32 * c], acc[c], 0, 0, 0);
      }
      acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x_[(k + h) * 33], acc[4], 0, 0, 0);
    }
    if (more) commit(buf ^ 1);               // the other buffer: its readers finished before the previous barrier
    __syncthreads();
  }
  // this slab's partial results: [G*H | G*o | G]
  float* sl = g.slab + (long)blockIdx.y * (G * H + G * o + G);
  if (!rev) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) sl[(long)(row0 + 32 * wave + drow16(i, h)) * H + 32 * c + r] = acc[c][i];
  }
  if (r < o) {
#pragma unroll
    for (int i = 0; i < 16; ++i) sl[(long)G * H + (long)(row0 + 32 * wave + drow16(i, h)) * o + r] = acc[4][i];
  }
  rs += __shfl_xor(rs, 32);
  if (h == 0) sl[(long)G * H + (long)G * o + row0 + 32 * wave + r] = rs;
}

// Sum over the slabs in slab order, added to the gradient tensors (bias_ih and bias_hh share theirs); the
// reverse direction's slabs follow the forward ones and have no recurrent part.
__global__ __launch_bounds__(256) void lstm_window_dw_reduce_kernel(WinDwGroup grp, int splits, int splits_r) {
  __shared__ float part[2][8][33];
  constexpr int H = 128, G = 512;
  const WinDwInst& g = grp.in[blockIdx.y];
  const int P = G * H + G * g.o + G;
  const int c = threadIdx.x & 31, grp8 = threadIdx.x >> 5, e = blockIdx.x * 32 + c;
  float s = 0.f, sr = 0.f;
  if (e < P) {
    for (int sp = grp8; sp < splits; sp += 8) s += g.slab[(long)sp * P + e];
    if (e >= G * H)
      for (int sp = splits + grp8; sp < splits + splits_r; sp += 8) sr += g.slab[(long)sp * P + e];
  }
  part[0][grp8][c] = s;
  part[1][grp8][c] = sr;
  __syncthreads();
  if (grp8 == 0 && e < P) {
    float t = 0.f, tr = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      t += part[0][k][c];
      tr += part[1][k][c];
    }
    if (e < G * H) g.g_hh[e] += t;
    else if (e < G * H + G * g.o) {
      g.g_ih[e - G * H] += t;
      g.g_ih_r[e - G * H] += tr;
    } else {
      const int b = e - G * H - G * g.o;
      g.g_bih[b] += t;
      g.g_bhh[b] += t;
      g.g_bih_r[b] += tr;
      g.g_bhh_r[b] += tr;
    }
  }
}

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
  const bool seq = seq_fits(d, n, H, T);
  if (seq) {
    // transposition, all T steps and the reverse step in one launch
    SeqFwdGroup sg;
    for (int i = 0; i < n; ++i) {
      const float* const* p = d[i].params;
      sg.in[i] = SeqFwdInst{d[i].obs, p[0], p[1], p[2], p[3], p[4], p[6], p[7], w[i].x_tm, w[i].gates, w[i].cbuf, w[i].hbuf,
                            w[i].cat, w[i].gates_r, w[i].c_r, d[i].ldobs, d[i].o, d[i].no_backward ? 0 : 1};
    }
    const int chunks = ceil_div(B, 4);
    // One round of the 256 CUs for the forward and the reverse workgroups of all modules (243 in all: a workgroup that
    // has to wait for a CU starts when a forward workgroup ends and doubles the launch).  A reverse workgroup runs
    // ONE step per group of four rows against T for a forward one, so a ninth of a module's workgroups keeps pace
    // (nine modules at batch 1024: 24 + 3 per module instead of 22 + 5, 225 -> 200 us).
    // Workgroups per module.  A module whose steps are not saved (no_backward) issues no stores inside its time loop, but
    // its steps are not measurably shorter (the stores drain under the products): on whole HATD3Triple updates at batch
    // 1024 equal shares were the fastest (shares 0.55 / 0.65 / 0.76 / 0.9 / 1.0 of a saving module's workgroups: 1.268 /
    // 1.245 / 1.220 / 1.208 / 1.212 ms).  HHRL_LSTM_NOSAVE_SHARE sets another share for sweeps.
    static const double nosave_w = [] { const char* e = getenv("HHRL_LSTM_NOSAVE_SHARE"); return e ? atof(e) : 1.0; }();
    double wsum = 0.0;
    for (int i = 0; i < n; ++i) wsum += sg.in[i].save ? 1.0 : nosave_w;
    sg.n = n;
    sg.wg0[0] = 0;
    for (int i = 0; i < n; ++i) {
      int per = (int)(243.0 * (sg.in[i].save ? 1.0 : nosave_w) / wsum);
      if (per < 2) per = 2;
      int gx_r = per / 9 > 1 ? per / 9 : 1;
      int gx_f = per - gx_r;
      if (gx_f > chunks) gx_f = chunks;
      if (gx_r > chunks) gx_r = chunks;
      sg.n_rev[i] = (short)gx_r;
      sg.wg0[i + 1] = (short)(sg.wg0[i] + gx_r + gx_f);
    }
    const dim3 grid(sg.wg0[n], 1, 1);
    const size_t lds_fwd = 4 * (size_t)(SQ_SLAB + T * 4 * 32);
    HHRL_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(lstm_seq_forward_kernel<16>), (int)lds_fwd, "lstm"));
    HHRL_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(lstm_seq_forward_kernel<32>), (int)lds_fwd, "lstm"));
    ProfScope ps("lstm_seq_forward_kernel", st, 2.0 * B * G * ((double)T * (H + omax) + omax) * n,
                 4.0 * n * ((double)B * T * (G + 2.0 * H + 2.0 * omax) + (double)G * (H + 2.0 * omax)));
#ifdef HHRL_STAMPS
    {
      const int first_fwd = sg.n_rev[0];
      (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_lstm_stamp_wg), &first_fwd, sizeof(int), 0, hipMemcpyHostToDevice, st);
    }
#endif
    if (omax <= 16) hipLaunchKernelGGL(lstm_seq_forward_kernel<16>, grid, dim3(512), lds_fwd, st, sg, B, T, chunks);
    else hipLaunchKernelGGL(lstm_seq_forward_kernel<32>, grid, dim3(512), lds_fwd, st, sg, B, T, chunks);
    HHRL_TRY(launch_status("lstm_seq_forward_kernel"));
  }
  if (!seq) {
    const long work = (long)B * T * omax > BH ? (long)B * T * omax : BH;
    ProfScope ps("lstm_prep_kernel", st, 0.0, 8.0 * B * T * omax * n);
    hipLaunchKernelGGL(lstm_prep_kernel, dim3(ceil_div(work, 256), 1, n), dim3(256), 0, st, pg, B, T, H);
    HHRL_TRY(launch_status("lstm_prep_kernel"));
  }
  const bool fused = !seq && fused_steps_fit(d, n, H);
  if (!fused && !seq) {
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
  const bool fused = !seq_fits(d, n, H, T) && fused_steps_fit(d, n, H);
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
    bool small = true;
    for (int i = 0; i < n; ++i) small = small && gemm_is_small(dx[i]) && gemm_is_small(dw[i]);
    if (small) {
      // one launch: input- and weight-gradient products side by side, the bias gradient as the row sums of the
      // weight-gradient product's operand
      GemmArgs all[2 * kMaxInst];
      for (int i = 0; i < n; ++i) {
        dw[i].rowsum = d[i].grads[9];
        all[2 * i] = dx[i];
        all[2 * i + 1] = dw[i];
      }
      for (int base = 0; base < 2 * n; base += kGemmGroupMax)
        HHRL_TRY(gemm_group(all + base, 2 * n - base < kGemmGroupMax ? 2 * n - base : kGemmGroupMax, st));
    } else {
      {
        ProfScope ps("colsum2_kernel", st, 0.0, 4.0 * B * H * n);
        hipLaunchKernelGGL(colsum2_kernel, dim3(ceil_div(H, 64), ceil_div(B, 64), n), dim3(256), 0, st, cg, d[0].lddo, H, 64);
        HHRL_TRY(launch_status("colsum2_kernel(fc)"));
      }
      HHRL_TRY(gemm_group(dx, n, st));
      HHRL_TRY(gemm_group(dw, n, st));
    }
  }
  const bool seq = seq_fits(d, n, H, T);
  if (seq) {
    // every cell of the window, both directions, in one launch
    SeqBwdGroup sg;
    for (int i = 0; i < n; ++i)
      sg.in[i] = SeqBwdInst{d[i].params[1], w[i].gates, w[i].cbuf, w[i].dcat, w[i].gates_r, w[i].c_r, w[i].dgates, w[i].dgates_r};
    const int chunks = ceil_div(B, 4);
    ProfScope ps("lstm_seq_backward_kernel", st, 2.0 * B * G * H * (double)(T - 1) * n,
                 4.0 * n * ((double)B * T * (2.0 * G + 2.0 * H) + (double)G * H));
    hipLaunchKernelGGL(lstm_seq_backward_kernel, dim3(seq_grid_x(chunks, n), 2, n), dim3(512), 0, st, sg, B, T, chunks);
    HHRL_TRY(launch_status("lstm_seq_backward_kernel"));
  }
  // cells whose hidden gradient comes from the fc layer: last forward step and the reverse step
  if (!seq) {
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
  for (int t = T - 1; t >= 1 && !fused && !seq; --t) {
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
    bool win = n <= kMaxInst;
    for (int i = 0; i < n; ++i) win = win && win_splits(B, T, d[i].o, H) > 0;
    GemmArgs dw[3 * kMaxInst];
    int m = 0;
    if (win) {
      // long reductions: the dedicated kernel for the forward direction's two products and the bias sums
      WinDwGroup wg;
      int splits = 64, omax = 0;
      for (int i = 0; i < n; ++i) {
        const int o = d[i].o;
        wg.in[i] = WinDwInst{w[i].dgates, w[i].hbuf, w[i].x_tm, w[i].dgates_r, w[i].x_tm + (long)(T - 1) * B * o, w[i].dwslab,
                             d[i].grads[0], d[i].grads[1], d[i].grads[2], d[i].grads[3], d[i].grads[4], d[i].grads[6], d[i].grads[7], o};
        const int sp = win_splits(B, T, o, H);
        splits = sp < splits ? sp : splits;
        omax = o > omax ? o : omax;
      }
      // about two workgroups per CU over the group (four row tiles per module and slab)
      while (splits > 4 && 4L * n * splits > 640) --splits;
      const int K = T * B;
      const int kslab = ceil_div(ceil_div(K, splits), kWinKc) * kWinKc;
      splits = ceil_div(K, kslab);
      int splits_r = win_splits_rev(B);
      const int kslab_r = ceil_div(ceil_div(B, splits_r), kWinKc) * kWinKc;
      splits_r = ceil_div(B, kslab_r);
      HHRL_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(lstm_window_dw_kernel), kWinLdsFloats * 4, "lstm_window_dw"));
      {
        ProfScope ps("lstm_window_dw_kernel", st, 2.0 * G * ((double)K * (H + omax) + (double)B * omax) * n,
                     4.0 * n * (K + (double)B) * (G + H + omax));
        hipLaunchKernelGGL(lstm_window_dw_kernel, dim3(G / 128, splits + splits_r, n), dim3(256), kWinLdsFloats * 4, st, wg, K, kslab,
                           splits, B, kslab_r);
        HHRL_TRY(launch_status("lstm_window_dw_kernel"));
      }
      {
        const int P = G * H + G * omax + G;
        ProfScope ps("lstm_window_dw_reduce_kernel", st, 0.0, 4.0 * n * (double)P * (splits + splits_r + 2));
        hipLaunchKernelGGL(lstm_window_dw_reduce_kernel, dim3(ceil_div(P, 32), n), dim3(256), 0, st, wg, splits, splits_r);
        HHRL_TRY(launch_status("lstm_window_dw_reduce_kernel"));
      }
    }
    for (int i = 0; i < n; ++i) {
      const int o = d[i].o;
      if (!win) {
        dw[m++] = gemm_args(w[i].dgates, 1, G, w[i].x_tm, o, 1, d[i].grads[0], o, G, o, T * B, nullptr, HHRL_ACT_NONE, 2);
        dw[m] = gemm_args(w[i].dgates, 1, G, w[i].hbuf, H, 1, d[i].grads[1], H, G, H, T * B, nullptr, HHRL_ACT_NONE, 2);
        // bias_ih and bias_hh share a gradient, the column sums of dgates: the row sums of this product's operand
        dw[m].rowsum = d[i].grads[2];
        dw[m++].rowsum2 = d[i].grads[3];
        dw[m] = gemm_args(w[i].dgates_r, 1, G, w[i].x_tm + (long)(T - 1) * B * o, o, 1, d[i].grads[4], o, G, o, B, nullptr,
                          HHRL_ACT_NONE, 2);
        dw[m].rowsum = d[i].grads[6];
        dw[m++].rowsum2 = d[i].grads[7];
      }
    }
    // (each of these has at most 16 x 4 output tiles: always the small-product kernels, which honour rowsum)
    for (int base = 0; base < m; base += kGemmGroupMax)
      HHRL_TRY(gemm_group(dw + base, m - base < kGemmGroupMax ? m - base : kGemmGroupMax, st));
  }
  return HHRL_OK;
}

#ifdef HHRL_STAMPS
int hhrl_debug_lstm_stamps(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lstm_stamps), sizeof(long long) * 2 * 64) == hipSuccess ? 0 : -1;
}
#endif

int hhrl_lstm_forward(const float* obs, int ldobs, const float* const* params, float* out, int ldo, float* ws, int B,
                      int T, int o, int H, void* stream) {
  hhrl_lstm_desc d;
  d.obs = obs; d.ldobs = ldobs; d.o = o;
  d.params = params; d.grads = nullptr;
  d.out = out; d.ldo = ldo;
  d.d_out = nullptr; d.lddo = 0;
  d.ws = ws;
  d.no_backward = 0;
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
  d.no_backward = 0;
  return hhrl_lstm_backward_group(&d, 1, B, T, H, stream);
}

}  // extern "C"
